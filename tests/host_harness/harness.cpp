// CPU-only test harness: exposes the engine's host-side helpers (pose math shared
// with the head kernels, 16-bit conversions, weight packers) to pytest via ctypes so
// they can be checked against the oracle without a GPU.  Not part of the product.
#include "host_pack.h"
#include "pose_math.h"

extern "C" {

void hh_procrustes(const float* M, float* R, int n) {
  for (int i = 0; i < n; ++i) procrustes3x3(M + 9 * i, R + 9 * i);
}

void hh_procrustes_jacobi(const float* M, float* R, int n) {
  for (int i = 0; i < n; ++i) procrustes3x3_jacobi(M + 9 * i, R + 9 * i);
}

// returns how many inputs took the Newton fast path
int hh_procrustes_newton_count(const float* M, int n) {
  int c = 0;
  float R[9];
  for (int i = 0; i < n; ++i) c += procrustes3x3_newton(M + 9 * i, R) ? 1 : 0;
  return c;
}

void hh_nullify_yaw(const float* R, float* O, int n) {
  for (int i = 0; i < n; ++i) nullify_yaw3x3(R + 9 * i, O + 9 * i);
}

void hh_cvt16(const float* src, uint16_t* dst, long n, int dtype) {
  for (long i = 0; i < n; ++i) dst[i] = flope_host::cvt16(src[i], dtype);
}

int hh_lds_row_to_channel(int rl) { return flope_host::lds_row_to_channel(rl); }
int hh_stag_row_to_channel(int rl) { return flope_host::stag_row_to_channel(rl); }

// packed conv image -> caller buffer (cout*cin*k*k uint16)
void hh_pack_conv(const float* w, int cout, int cin, int k, int dtype, uint16_t* dst) {
  std::vector<float> wf(w, w + (size_t)cout * cin * k * k);
  std::vector<uint16_t> p = flope_host::pack_conv(wf, cout, cin, k, dtype);
  memcpy(dst, p.data(), p.size() * 2);
}

void hh_pack_stem(const float* w, int dtype, uint16_t* dst) {
  std::vector<float> wf(w, w + (size_t)64 * 3 * 7 * 7);
  std::vector<uint16_t> p = flope_host::pack_stem(wf, dtype);
  memcpy(dst, p.data(), p.size() * 2);
}

}  // extern "C"
