// probe: layout and broadcast semantics of v_mfma_f32_4x4x1_16B_f32 (cbsz / abid) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__global__ void k(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, 0);
  for (int r = 0; r < 4; ++r) d[r * 64 + l] = c[r];
}
template <int CBSZ, int ABID>
void run(const char* name) {
  float ha[64], hb[64], hd[256];
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f * (1 + l); }   // product identifies (a lane, b lane)
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  printf("%s\n", name);
  for (int l = 0; l < 64; l += 1) {
    if (!(l < 8 || (l >= 32 && l < 40) || l >= 60)) continue;
    printf(" lane %2d:", l);
    for (int r = 0; r < 4; ++r) {
      const float v = hd[r * 64 + l];
      // decode: v = (1 + la) * 100 * (1 + lb)
      int la = -1, lb = -1;
      for (int x = 0; x < 64 && la < 0; ++x) for (int y = 0; y < 64; ++y) if (v == (1.0f + x) * 100.0f * (1 + y) && (y == l)) { la = x; lb = y; break; }
      printf("  r%d: a[%2d]*b[%2d]", r, la, lb);
    }
    printf("\n");
  }
}
int main() {
  run<0, 0>("cbsz=0 abid=0");
  run<3, 0>("cbsz=3 abid=0");
  run<3, 5>("cbsz=3 abid=5");
  run<4, 9>("cbsz=4 abid=9");
  run<2, 1>("cbsz=2 abid=1");
  return 0;
}
