// GPU check of the shared-divisor quotients of csrc/rtc_kernels.hip (refined_rcp / quotient) against the plain
// division x / d, bit for bit, over random and edge-case operands inside the range cube_slab() uses them in.
// Built and run by tests/test_parity_gpu.py::test_shared_divisor_quotients_are_exact.  TEST INFRASTRUCTURE.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ double refined_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double quotient(double x, double d, double r) {
  const double q = x * r;
  return __builtin_fma(__builtin_fma(-d, q, x), r, q);
}

__device__ uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// mode 0: origin and direction as cube_slab sees them: |d| in [1e-5, 2^200], numerators -1 - o and 1 - o, |o| <= 2^200
// mode 1: |d| close to the 1e-5 threshold, origins close to +-1 (numerators of a few ulps, and exact zeros)
__global__ void check(uint64_t seed, int mode, unsigned long long* mismatches, double* first) {
  const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
  uint64_t a = mix(seed + 2 * i), b = mix(seed + 2 * i + 1);
  double d, o;
  if (mode == 0) {
    const int de = static_cast<int>(a % 217u) - 16;            // 2^-16 .. 2^200
    d = ldexp(1.0 + static_cast<double>((a >> 11) & 0xFFFFFFFFFFFFFull) * 0x1p-52, de);
    if (fabs(d) < 1e-5) d = 1e-5;
    if ((a >> 8) & 1) d = -d;
    const int oe = static_cast<int>(b % 260u) - 60;            // 2^-60 .. 2^199
    o = ldexp(1.0 + static_cast<double>((b >> 11) & 0xFFFFFFFFFFFFFull) * 0x1p-52, oe);
    if ((b >> 8) & 1) o = -o;
  } else {
    d = 1e-5 * (1.0 + static_cast<double>(a & 0xFFFF) * 0x1p-52);
    if ((a >> 20) & 1) d = -d;
    if ((a >> 21) & 1) d = ldexp(d, static_cast<int>((a >> 22) % 40u));
    o = ((b >> 1) & 1 ? 1.0 : -1.0) + (static_cast<double>((b >> 8) & 0xFF) - 128.0) * 0x1p-53;
    if ((b >> 2) & 3) o = ((b >> 1) & 1 ? 1.0 : -1.0) * (1.0 + static_cast<double>((b >> 8) & 0xFFFF) * 0x1p-52);
  }
  const double n0 = -1.0 - o, n1 = 1.0 - o;
  const double r = refined_rcp(d);
  const double q0 = quotient(n0, d, r), q1 = quotient(n1, d, r);
  const double p0 = n0 / d, p1 = n1 / d;
  if (__builtin_bit_cast(uint64_t, q0) != __builtin_bit_cast(uint64_t, p0) ||
      __builtin_bit_cast(uint64_t, q1) != __builtin_bit_cast(uint64_t, p1)) {
    if (atomicAdd(mismatches, 1ull) == 0ull) {
      first[0] = d;
      first[1] = o;
      first[2] = q0;
      first[3] = p0;
      first[4] = q1;
      first[5] = p1;
    }
  }
}

int main() {
  unsigned long long* d_bad;
  double* d_first;
  (void)hipMalloc(&d_bad, sizeof *d_bad);
  (void)hipMalloc(&d_first, 6 * sizeof(double));
  (void)hipMemset(d_bad, 0, sizeof *d_bad);
  unsigned long long total = 0;
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 8; ++rep) {
      hipLaunchKernelGGL(check, dim3(1 << 16), dim3(256), 0, 0, 0x1234567ull * (rep + 1) + mode, mode, d_bad, d_first);
      total += (1ull << 24);
    }
  if (hipDeviceSynchronize() != hipSuccess) {
    std::printf("shared_divisor_check: HIP error\n");
    return 2;
  }
  unsigned long long bad = 0;
  double first[6];
  (void)hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost);
  (void)hipMemcpy(first, d_first, sizeof first, hipMemcpyDeviceToHost);
  std::printf("shared_divisor_check: %llu operand pairs, %llu mismatches\n", total, bad);
  if (bad) std::printf("first: d=%a o=%a  shared %a plain %a | shared %a plain %a\n", first[0], first[1], first[2], first[3], first[4], first[5]);
  return bad ? 1 : 0;
}
