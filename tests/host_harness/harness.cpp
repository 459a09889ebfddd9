// CPU-only test harness: exposes the engine's host-side helpers (pose math shared
// with the head kernels, 16-bit conversions, weight packers) to pytest via ctypes so
// they can be checked against the oracle without a GPU.  Not part of the product.
#include "host_pack.h"
#include "pose_math.h"
#include "w4_sched.h"

#include <cstdio>
#include <vector>

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

// register-weight fragment images of conv_s2r (64 -> 128, stride 2) and conv_s1r (128 -> 128, K split over wave pairs), r05
void hh_pack_s2r(const float* w, int dtype, uint16_t* dst) {
  std::vector<float> wf(w, w + (size_t)128 * 64 * 9);
  std::vector<uint16_t> p = flope_host::pack_s2r(wf, 128, 64, dtype);
  memcpy(dst, p.data(), p.size() * 2);
}

void hh_pack_s1r(const float* w, int dtype, uint16_t* dst) {
  std::vector<float> wf(w, w + (size_t)128 * 128 * 9);
  std::vector<uint16_t> p = flope_host::pack_s1r(wf, 128, dtype);
  memcpy(dst, p.data(), p.size() * 2);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// conv_w4's hand-counted waits (flope_amd/csrc/w4_sched.h), checked against an INDEPENDENT statement of the schedule: the
// per-wave sequence of DMA issues (which LDS slot, which contents), `s_waitcnt vmcnt(N)` waits, barriers and LDS reads of one body
// of the K loop is replayed operation by operation, and every read must find its bytes landed -- the producing DMA complete in
// the issuing wave (vmcnt leaves only the N YOUNGEST operations in flight) before the barrier the reader has passed -- and no
// DMA may be issued into a slot before the barrier behind the last read of its old contents.
//   nbd      weight-ring depth (3..5);  pw  pieces of a patch burst;  spread  parts of the cold burst (w4_sched.h);  mt  pixel tiles
//   boundary 1: a class walk's tile boundary lies in front of the checked body (2 mt epilogue stores in the queue)
//   res      1: the checked body is a class walk's last body with a residual input (register loads in sub-steps 14 and 16)
//   formula  0: w4_wait_n (what the kernel uses);  1: r03's closed form;  2: the form r03 replaced (one burst too lax at barriers 3
//            and 8 of a 5-deep ring) -- 1 and 2 only describe spread <= 1, boundary = res = 0
// Returns 0 when every read is covered, else a positive code; msg names the first violation.
// tgw: LDS-DMA pieces per wave and double tile -- 4 = conv_w4 (four waves), 2 = conv_w8 (eight waves, pw = PT pieces per burst)
extern "C" int flope_host_w4_schedule_check_tgw(int nbd, int pw, int spread, int mt, int boundary, int res, int formula, int tgw, char* msg, int msg_cap);
extern "C" int flope_host_w4_schedule_check(int nbd, int pw, int spread, int mt, int boundary, int res, int formula, char* msg, int msg_cap) {
  return flope_host_w4_schedule_check_tgw(nbd, pw, spread, mt, boundary, res, formula, W4_TGW, msg, msg_cap);
}
extern "C" int flope_host_w4_schedule_check_tgw(int nbd, int pw, int spread, int mt, int boundary, int res, int formula, int tgw, char* msg, int msg_cap) {
  struct Op { int kind, a, b; };                         // kind 0: double tile a (absolute index, body * 9 + k) piece b; 1: patch buffer a, read in body b; 2: store; 3: residual load
  auto say = [&](const char* fmt, int x, int y, int z) { if (msg && msg_cap > 0) snprintf(msg, msg_cap, fmt, x, y, z); };
  if (nbd < 3 || nbd > 5 || pw < 2 || mt < 4 || mt > 8 || tgw < 1 || tgw > 8 || (formula != 0 && tgw != W4_TGW)) { say("bad arguments", 0, 0, 0); return 100; }
  const int PD = nbd - 1;
  std::vector<Op> q;                                      // this wave's memory operations in issue order
  auto issue_second = [&](int body, int d) {              // second sub-step of double step d of `body` (behind its barrier d)
    for (int i = 0; i < tgw; ++i) q.push_back({0, body * 9 + d + PD, i});
    const int n = w4_patch_pieces(d, pw, spread), buf = w4_patch_buffer(d, pw, spread);
    for (int i = 0; i < n; ++i) q.push_back({1, buf, buf == 1 ? body : body + 1});
  };
  // WAR, stated once: the ring slot of double tile D + PD held double tile D - 1 (nbd = PD + 1 slots), last read in the first sub-step
  // of D - 1, i.e. before barrier D - 1 < D; patch buffer 1 is last read in the first sub-step of D = 8 (fragments of sub-step 17),
  // so a burst into it may be issued from behind barrier 8 on, i.e. at D = 8 or in the next body; buffer 0 is last read in the second
  // sub-step of D = 3 (fragments of sub-step 8), so pieces for it may be issued from behind barrier 4 on.
  for (int d = 0; d < 9; ++d) {
    const int buf = w4_patch_buffer(d, pw, spread);
    if (buf == 0 && d < 4) { say("patch pieces for buffer 0 issued at double step %d, before barrier 4", d, 0, 0); return 1; }
    if (buf == 1 && d != 0 && d != 8) { say("patch pieces for buffer 1 issued at double step %d", d, 0, 0); return 2; }
  }
  {
    int total = 0;
    for (int d = 1; d < 9; ++d) total += w4_patch_pieces(d, pw, spread);
    if (total != pw || w4_patch_pieces(0, pw, spread) != pw) { say("a burst has %d pieces instead of %d", total, pw, 0); return 3; }
  }
  for (int d = 0; d < 9; ++d) issue_second(-1, d);        // the body before (its own reads are not checked here)
  if (boundary) for (int i = 0; i < 2 * mt; ++i) q.push_back({2, 0, 0});
  for (int D = 0; D < 9; ++D) {
    if (res) for (int i = 0; i < w4_res_loads(D, mt); ++i) q.push_back({3, 0, 0});   // first sub-step of D
    // the wait in front of barrier D
    int N;
    if (formula == 0) N = w4_wait_n(D, PD, pw, spread, boundary ? 2 * mt : 0, res ? mt : 0, tgw);
    else if (formula == 1) N = w4_wait_n_r03(D, PD, pw);
    else N = w4_wait_n_r03_lax(D, PD, pw);
    const int landed = (int)q.size() - N;                 // operations 0 .. landed - 1 are complete in this wave (every wave runs the same stream)
    auto complete = [&](int kind, int a, int b) {         // every piece of that object complete?  (it must have been issued at all)
      int seen = 0;
      for (int i = 0; i < (int)q.size(); ++i)
        if (q[i].kind == kind && q[i].a == a && (kind == 0 || q[i].b == b)) { ++seen; if (i >= landed) return false; }
      return seen == (kind == 0 ? tgw : pw);
    };
    // reads behind barrier D (second sub-step of D, first sub-step of D + 1): double tile D + 1; fragments of sub-steps 2 D + 2 and
    // 2 D + 3 from patch buffer (sub-step / 9) -- sub-step >= 18 is the next body's buffer 0
    if (D + 1 <= 8 + PD && !complete(0, D + 1, 0)) { say("barrier %d: double tile %d read before its DMA is covered (vmcnt %d)", D, D + 1, N); return 10 + D; }
    for (int u = 2 * D + 2; u <= 2 * D + 3; ++u) {
      if (u > 18) continue;                               // (sub-step 19 belongs to the next body's barrier 0)
      const int buf = u >= 18 ? 0 : u / 9, body = u >= 18 ? 1 : 0;
      if (!complete(1, buf, body)) { say("barrier %d: patch buffer %d read (sub-step %d) before its burst is covered", D, buf, u); return 30 + D; }
    }
    issue_second(0, D);
  }
  say("ok", 0, 0, 0);
  return 0;
}
