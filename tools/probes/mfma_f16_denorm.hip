// Probe: does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs, and what is the A/B lane->k mapping?
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16* A, const _Float16* B, float* D) {  // A[32][16] row-major, B[16][32] row-major
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = acc[i]; }
}
int main() {
  std::vector<_Float16> A(32 * 16), B(16 * 32);
  std::vector<float> D(32 * 32), ref(32 * 32, 0.f);
  // test 1: mapping with asymmetric integer data
  for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 16; ++kk) A[i * 16 + kk] = (_Float16)((i * 3 + kk * 7) % 11 - 5);
  for (int kk = 0; kk < 16; ++kk) for (int j = 0; j < 32; ++j) B[kk * 32 + j] = (_Float16)((kk * 5 + j * 2) % 13 - 6);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int kk = 0; kk < 16; ++kk) ref[i * 32 + j] += (float)A[i * 16 + kk] * (float)B[kk * 32 + j];
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(D[i] - ref[i]));
  printf("mapping max err %g\n", err);
  // test 2: subnormal A (2^-20 = 9.5e-7 < 6.1e-5 normal min) times B = 1024
  for (auto& v : A) v = (_Float16)0; for (auto& v : B) v = (_Float16)0;
  A[0] = (_Float16)9.5367431640625e-07f; B[0] = (_Float16)1024.0f;         // D[0][0] should be 2^-10 = 9.765625e-4
  A[1 * 16 + 1] = (_Float16)1024.0f; B[1 * 32 + 1] = (_Float16)5.9604645e-08f;  // smallest subnormal as B: D[1][1] = 6.1e-5
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  printf("subnormal A: got %g expect %g ; subnormal B: got %g expect %g\n", D[0], 9.765625e-4, D[33], 1024.0 * 5.9604645e-08);
  return 0;
}
