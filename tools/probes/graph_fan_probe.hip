// Minimal stand-alone form of the hipGraph behaviour round 5 met inside the engine (docs/LAB_NOTEBOOK.md section 18, profiles/r05_fork.txt): a captured
// graph in which ~20 side-stream nodes each carry their OWN (redundant) edge from the SAME chain node ran the chain node behind that fan before
// its predecessor had finished -- it read the previous replay's operands.  Here: a chain m0 .. m9 on the main stream; 20 side kernels, each
// preceded by hipEventRecord(main) + hipStreamWaitEvent(side) although the main stream has not moved (20 edges m9 -> s_j); m10 .. m19 on the
// main stream; join; m20.  Every kernel stamps {start, end} with the 100 MHz wall clock and the chain carries a value (m_i: d[i] = d[i-1] + 1
// after a delay), so an early node shows twice: start(m_i) < end(m_{i-1}), and d[19] != replay id + 19.  FAN=1 captures the one-edge form the
// engine uses now (the side stream waits only when the main stream has moved).
//   hipcc --offload-arch=gfx950 -O2 -o graph_fan_probe.bin graph_fan_probe.hip && ./graph_fan_probe.bin [replays] [fan: 20|1] [delay_us]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void node(long long* ts, int idx, const int* din, int* dout, int init, int delay_ticks) {
  const long long t0 = wall_clock64();
  int v = din ? *(volatile const int*)din : init;
  while (wall_clock64() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(8);
  if (dout) *dout = v + 1;
  __threadfence();
  ts[2 * idx] = t0; ts[2 * idx + 1] = wall_clock64();
}

int main(int argc, char** argv) {
  const int replays = argc > 1 ? atoi(argv[1]) : 200, fan = argc > 2 ? atoi(argv[2]) : 20, delay_us = argc > 3 ? atoi(argv[3]) : 20;
  const int NM = 21, NS = 20, delay = delay_us * 100;
  int rv = 0; CK(hipRuntimeGetVersion(&rv));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s, HIP runtime %d, %d replays, %s, node delay %d us\n", pr.gcnArchName, rv, replays, fan > 1 ? "one redundant edge per side node (fan)" : "one edge per fork", delay_us);
  long long* ts; int* d; int* seed;
  CK(hipMalloc(&ts, sizeof(long long) * 2 * (NM + NS))); CK(hipMalloc(&d, sizeof(int) * (NM + 1))); CK(hipMalloc(&seed, sizeof(int)));
  CK(hipMemset(d, 0, sizeof(int) * (NM + 1)));
  hipStream_t mainS, sideS; CK(hipStreamCreate(&mainS)); CK(hipStreamCreate(&sideS));
  std::vector<hipEvent_t> ev(NS + 2);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(mainS, hipStreamCaptureModeThreadLocal));
  node<<<1, 64, 0, mainS>>>(ts, 0, seed, d + 0, 0, delay);                                   // m0: d[0] = seed + 1
  for (int i = 1; i < 10; ++i) node<<<1, 64, 0, mainS>>>(ts, i, d + i - 1, d + i, 0, delay);
  for (int j = 0; j < NS; ++j) {
    if (fan > 1 || j == 0) { CK(hipEventRecord(ev[j], mainS)); CK(hipStreamWaitEvent(sideS, ev[j], 0)); }
    node<<<1, 64, 0, sideS>>>(ts, NM + j, d + 9, nullptr, 0, 3 * delay);                      // side: reads what m9 wrote
  }
  for (int i = 10; i < 20; ++i) node<<<1, 64, 0, mainS>>>(ts, i, d + i - 1, d + i, 0, delay);
  CK(hipEventRecord(ev[NS], sideS)); CK(hipStreamWaitEvent(mainS, ev[NS], 0));
  node<<<1, 64, 0, mainS>>>(ts, 20, d + 19, d + 20, 0, delay);
  CK(hipStreamEndCapture(mainS, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
  size_t ne = 0; CK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
  printf("graph: %zu nodes, %zu edges\n", nn, ne);
  std::vector<long long> h(2 * (NM + NS)); std::vector<int> hd(NM + 1);
  int bad_order = 0, bad_value = 0, first_bad = -1;
  for (int r = 0; r < replays; ++r) {
    int s = 1000 * (r + 1);
    CK(hipMemcpyAsync(seed, &s, sizeof(int), hipMemcpyHostToDevice, mainS));
    CK(hipGraphLaunch(ge, mainS)); CK(hipStreamSynchronize(mainS));
    CK(hipMemcpy(h.data(), ts, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hd.data(), d, sizeof(int) * hd.size(), hipMemcpyDeviceToHost));
    bool bo = false;
    for (int i = 1; i < NM; ++i) if (h[2 * i] < h[2 * (i - 1) + 1]) { bo = true; if (first_bad < 0) { first_bad = i; printf("replay %d: m%d started %lld ticks BEFORE m%d ended\n", r, i, h[2 * (i - 1) + 1] - h[2 * i], i - 1); } }
    for (int j = 0; j < NS; ++j) if (h[2 * (NM + j)] < h[2 * 9 + 1]) { bo = true; if (first_bad < 0) { first_bad = 100 + j; printf("replay %d: side node %d started before m9 ended\n", r, j); } }
    if (h[2 * 20] < h[2 * (NM + NS - 1) + 1]) { bo = true; if (first_bad < 0) { first_bad = 200; printf("replay %d: m20 started before the last side node ended\n", r); } }
    bad_order += bo;
    bad_value += hd[20] != s + 21;
  }
  printf("replays with a node that started before a predecessor ended: %d of %d; with a wrong chain value: %d of %d\n", bad_order, replays, bad_value, replays);
  return 0;
}
