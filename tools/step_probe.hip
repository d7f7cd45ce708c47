// Standalone timing probe for the step kernel (development tool, not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I avsr-tf1_amd/csrc [-DPROBE_...] tools/step_probe.hip -o /tmp/probe
// Builds C4-like forward / backward launches (B=64, H=256, 3 wavefront tasks) and reports us per launch.
#include <cstdio>
#include <vector>
#include "../avsr-tf1_amd/csrc/capi.hip"
#include "../avsr-tf1_amd/csrc/step.hip"

using namespace avsr;
static float* dalloc(size_t n, float v = 0.01f) {
  float* p; hipMalloc(&p, n * sizeof(float));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = v * (float)((i * 2654435761u) % 1000) / 1000.f - v * 0.5f;
  hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return p;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, H = 256, T = 64, NT = argc > 2 ? atoi(argv[2]) : 3;
  StepLaunch F{}, Bk{};
  // forward: task0 hoisted (K=H), others K=2H
  for (int i = 0; i < NT; ++i) {
    StepTask& tk = F.task[i];
    const int in = (i == 0) ? 0 : H;
    float* wt = dalloc((size_t)4 * H * (H + H));
    if (in) { tk.src[tk.nsrc++] = StepSrc{dalloc((size_t)B * H), wt, H, 2 * H, H, SRC_PLAIN}; }
    tk.src[tk.nsrc++] = StepSrc{dalloc((size_t)B * H), wt + H, H, 2 * H, H, SRC_PLAIN};
    tk.B = B; tk.N = 4 * H; tk.mode = EP_LSTM_FWD; tk.t = 3; tk.T = T; tk.bias = dalloc(4 * H);
    tk.p0 = dalloc((size_t)B * T * 4 * H); tk.p1 = dalloc((size_t)B * T * H); tk.p2 = dalloc((size_t)B * (T + 2) * H);
    tk.s0 = (long)(T + 2) * H; tk.s1 = H; tk.s2 = (i == 0);
    tk.p3 = dalloc((size_t)B * H); tk.p4 = dalloc((size_t)B * H); tk.p5 = dalloc((size_t)B * H); tk.p6 = dalloc((size_t)B * H);
  }
  F.ntask = NT;
  for (int i = 0; i < NT; ++i) {
    StepTask& tk = Bk.task[i];
    float* w = dalloc((size_t)2 * H * 4 * H);
    tk.src[tk.nsrc++] = StepSrc{dalloc((size_t)B * 4 * H), w + (size_t)H * 4 * H, 4 * H, 4 * H, 4 * H, SRC_PLAIN};
    if (i + 1 < NT) tk.src[tk.nsrc++] = StepSrc{dalloc((size_t)B * 4 * H), w, 4 * H, 4 * H, 4 * H, SRC_PLAIN};
    tk.B = B; tk.N = H; tk.mode = EP_LSTM_BWD; tk.t = 3; tk.T = T;
    tk.p0 = dalloc((size_t)B * T * 4 * H, 1.0f); tk.p1 = dalloc((size_t)B * T * H); tk.p2 = dalloc((size_t)B * T * 4 * H);
    tk.p3 = dalloc((size_t)B * 4 * H); tk.p4 = dalloc((size_t)B * H); tk.p5 = dalloc((size_t)B * H);
    tk.p6 = dalloc((size_t)B * H); tk.p7 = dalloc((size_t)B * H);
  }
  Bk.ntask = NT;
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int geo = argc > 3 ? atoi(argv[3]) : -1;
  avsr_step_set_geometry(geo);
  for (int which = 0; which < 2; ++which) {
    StepLaunch& L = which ? Bk : F;
    for (int i = 0; i < 20; ++i) avsr_step_launch_raw(&L, s);
    hipStreamSynchronize(s);
    const int N = 400;
    hipEventRecord(e0, s);
    for (int i = 0; i < N; ++i) avsr_step_launch_raw(&L, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // graph version (no host launch cost)
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) avsr_step_launch_raw(&L, s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float msg; hipEventElapsedTime(&msg, e0, e1);
    printf("geo=%d %s B=%d tasks=%d: eager %.2f us/launch, graph %.2f us/launch\n", geo, which ? "bwd" : "fwd", B, NT, 1e3 * ms / N, 1e3 * msg / N);
  }
  return 0;
}
