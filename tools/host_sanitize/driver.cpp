// Host-side sanitizer driver (round-2 review, "Missing 6": the ABI's host code -- argument checks, the hiprtc source
// generators of csrc/jit.hip, the side-job and serial-chain recorders, the program validators -- had never run under
// ASan / UBSan).  Linked against a HOST-ONLY, sanitizer-instrumented build of csrc/*.hip (tools/host_sanitize/run.sh;
// no device code, no GPU needed): every call below either succeeds on the host (validators, generators, dry-run
// compiles) or must come back with a non-zero status and a message -- never crash, never trip a sanitizer.
#include "../../include/henbun_hip.h"
#include <cstdio>
#include <cstring>
#include <vector>

static int fails = 0;
#define EXPECT(cond, what)                                   \
  do {                                                       \
    if (!(cond)) {                                           \
      ++fails;                                               \
      std::printf("FAIL %s (%s)\n", what, hb_last_error_string()); \
    }                                                        \
  } while (0)

int main() {
  EXPECT(hb_version() == HB_ABI_VERSION, "hb_version");
  float dummy[64] = {0};
  float* fp = dummy;   // a host pointer standing in for device memory: nothing below dereferences it on the host
  // ---- program validator (host only)
  {
    std::vector<char> image((size_t)hb_ewise_prog_image_bytes());
    long n = 0;
    int red = 0;
    const int code[2][5] = {{HB_EW_MUL, 2, 0, 1, 0}, {HB_EW_EXP, 3, 2, 0, 0}};
    const double params[2][2] = {{0, 0}, {0, 0}};
    const void* in[2] = {fp, fp};
    const long istr[2][2] = {{8, 1}, {0, 1}};
    void* out[1] = {fp};
    const int oreg[1] = {3};
    const long ostr[1][2] = {{8, 1}};
    const long shape[2] = {4, 8};
    EXPECT(hb_ewise_prog_build(2, &code[0][0], &params[0][0], 2, in, &istr[0][0], 1, out, oreg, &ostr[0][0], 2, shape, image.data(), &n,
                               &red) == 0 && n == 32,
           "hb_ewise_prog_build valid");
    const int bad[1][5] = {{HB_EW_EXP, 3, 999, 0, 0}};   // a register index past the register file
    EXPECT(hb_ewise_prog_build(1, &bad[0][0], &params[0][0], 2, in, &istr[0][0], 1, out, oreg, &ostr[0][0], 2, shape, image.data(), &n,
                               &red) != 0,
           "hb_ewise_prog_build rejects a register index out of range");
    EXPECT(hb_ewise_prog_build(1000, &code[0][0], &params[0][0], 2, in, &istr[0][0], 1, out, oreg, &ostr[0][0], 2, shape, image.data(), &n,
                               &red) != 0,
           "hb_ewise_prog_build rejects 1000 instructions");
    // compiled forms: source generation + hiprtc dry run (no device)
    if (hb_ewise_jit_available()) {
      char src[8192];
      EXPECT(hb_ewise_jit_build_f32(2, &code[0][0], &params[0][0], 2, in, &istr[0][0], 1, out, oreg, &ostr[0][0], 2, shape, nullptr, &n, &red,
                                    src, sizeof(src)) == 0 && std::strstr(src, "hb_jit_kernel"),
             "hb_ewise_jit_build_f32 dry run");
      EXPECT(hb_ewise_jit_build_f64(2, &code[0][0], &params[0][0], 2, in, &istr[0][0], 1, out, oreg, &ostr[0][0], 2, shape, nullptr, &n, &red,
                                    src, 16) == 0,
             "hb_ewise_jit_build_f64 dry run, truncated source buffer");
      const int ccode[3][5] = {{HB_COLPROG_MAX, 1, 0, -1, -1}, {HB_EW_SUB, 2, 0, 1, -1}, {HB_COLPROG_SUM, 3, 2, -1, -1}};
      const double cpar[3][2] = {{0, 0}, {0, 0}, {0, 0}};
      const void* cin[1] = {nullptr};
      const long cistr[1][2] = {{1000, 1}};
      void* cout[2] = {nullptr, nullptr};
      const int coreg[2] = {2, 3};
      const long costr[2][2] = {{1000, 1}, {0, 1}};
      EXPECT(hb_ewise_colprog_build_f32(3, &ccode[0][0], &cpar[0][0], 1, cin, &cistr[0][0], 2, cout, coreg, &costr[0][0], 4, 1000, nullptr, src,
                                        sizeof(src)) == 0,
             "hb_ewise_colprog_build_f32 dry run");
      EXPECT(hb_ewise_colprog_build_f64(3, &ccode[0][0], &cpar[0][0], 1, cin, &cistr[0][0], 2, cout, coreg, &costr[0][0], 99, 1000, nullptr, src,
                                        sizeof(src)) != 0,
             "hb_ewise_colprog_build rejects 99 rows");
      const int cbad[1][5] = {{HB_EW_EXP, 3, 2, -1, -1}};
      EXPECT(hb_ewise_colprog_build_f32(1, &cbad[0][0], &cpar[0][0], 1, cin, &cistr[0][0], 1, cout, coreg + 1, &costr[0][0], 4, 1000, nullptr, src,
                                        sizeof(src)) != 0,
             "hb_ewise_colprog_build rejects an undefined register");
      // serial-chain recorder: record two jobs, look at the source, compile dry, discard
      EXPECT(hb_chain_begin() == 0, "hb_chain_begin");
      long t = 0;
      EXPECT(hb_adam_step_f32(fp, fp, fp, fp, 64, 1e-3, 0.9, 0.999, 1e-8, 1.0, &t, 1, nullptr, 0, nullptr, nullptr, nullptr) == 0,
             "hb_adam_step recorded into a chain");
      EXPECT(hb_gauss_ll_fold_f32(fp, 8, fp, fp, fp, nullptr) == 0, "hb_gauss_ll_fold recorded into a chain");
      std::vector<char> big(1 << 16);
      EXPECT(hb_chain_source(big.data(), (long)big.size()) == 0 && std::strstr(big.data(), "hb_gauss_fold_body"), "hb_chain_source");
      EXPECT(hb_chain_source(big.data(), 8) == 0, "hb_chain_source, tiny buffer");
      EXPECT(hb_chain_compile_dry() == 0, "hb_chain_compile_dry");
      EXPECT(hb_chain_begin() == 0 && hb_chain_discard() == 0, "hb_chain_discard");
    } else {
      std::printf("note: hiprtc not available, generator checks skipped\n");
    }
  }
  // ---- side-job list: nothing pending, discard is harmless
  EXPECT(hb_side_pending() == 0 && hb_side_discard() == 0 && hb_side_pending() == 0, "hb_side_pending / hb_side_discard");
  // ---- shape helpers
  EXPECT(hb_sgp_ws_elems(1, 8192, 512, 1, 1) > 0, "hb_sgp_ws_elems");
  EXPECT(hb_sgp_strip_path(1, 8192, 512, 1, 1, HB_PREC_NATIVE) == 1 && hb_sgp_strip_path(1, 8192, 1024, 1, 1, HB_PREC_NATIVE) == 0,
         "hb_sgp_strip_path");
  EXPECT(hb_sgp_head_units(1, 8192, 512, 1, 1, HB_PREC_NATIVE, 1, 0, 0) == 256 && hb_sgp_head_units(1, 8192, 512, 1, 2, HB_PREC_NATIVE, 1, 0, 0) == 0,
         "hb_sgp_head_units");
  // ---- argument checks: each must return an error before touching the device
  int info = 0;
  EXPECT(hb_matmul_f32(nullptr, fp, fp, 1, 4, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, 1.0, 0.0, nullptr, 0, HB_ACT_NONE, 0, nullptr, 0, nullptr) != 0,
         "hb_matmul NULL operand");
  EXPECT(hb_matmul_f32(fp, fp, fp, 1, -4, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, 1.0, 0.0, nullptr, 0, HB_ACT_NONE, 0, nullptr, 0, nullptr) != 0,
         "hb_matmul negative extent");
  EXPECT(hb_matmul_f32(fp, fp, fp, 1, 4, 4, 4, 2, 4, 4, 0, 0, 0, 0, 0, 1.0, 0.0, nullptr, 0, HB_ACT_NONE, 0, nullptr, 0, nullptr) != 0,
         "hb_matmul leading dimension too small");
  EXPECT(hb_matmul_f32(fp, fp, fp, 1, 4, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, 1.0, 0.0, nullptr, 0, 99, 0, nullptr, 0, nullptr) != 0,
         "hb_matmul unknown activation");
  EXPECT(hb_cholesky_f32(nullptr, fp, 1, 64, &info, nullptr) != 0, "hb_cholesky NULL");
  EXPECT(hb_cholesky_f32(fp, fp, 1, 64, &info, nullptr) != 0, "hb_cholesky aliasing operands");
  EXPECT(hb_cholesky_inverse_f32(fp, fp + 1, nullptr, 1, 64, &info, fp, nullptr, 0, nullptr) != 0, "hb_cholesky_inverse NULL W");
  EXPECT(hb_sgp_fwd_f32(HB_KERN_RBF, 7, fp, 0, fp, fp, 1, fp, nullptr, HB_PREC_NATIVE, fp, fp, nullptr, 0, fp, fp, nullptr, fp, fp, 1, 64, 64, 1,
                        1, fp, nullptr) != 0,
         "hb_sgp_fwd unknown mode");
  EXPECT(hb_sgp_fwd_f32(HB_KERN_RBF, HB_SGP_DIAGONAL, fp, 0, fp, fp, 3, fp, nullptr, HB_PREC_NATIVE, fp, fp, nullptr, 0, fp, fp, nullptr, fp, fp,
                        1, 64, 64, 2, 1, fp, nullptr) != 0,
         "hb_sgp_fwd lengthscale count");
  EXPECT(hb_sgp_fwd_gauss_f32(HB_KERN_RBF, HB_SGP_DIAGONAL, fp, 0, fp, fp, 1, fp, fp, HB_PREC_NATIVE, fp, fp, nullptr, 0, fp, fp, nullptr, fp, fp,
                              1, 64, 64, 1, 1, fp, nullptr, nullptr, fp, 1.0, fp, nullptr, fp, 2, nullptr) != 0,
         "hb_sgp_fwd_gauss NULL y");
  long tt = 0;
  EXPECT(hb_adam_step_f32(nullptr, fp, fp, fp, 8, 1e-3, 0.9, 0.999, 1e-8, 1.0, &tt, 1, nullptr, 0, nullptr, nullptr, nullptr) != 0,
         "hb_adam_step NULL theta");
  EXPECT(hb_adam_step_f32(fp, fp, fp, fp, 8, 1e-3, 0.9, 0.999, 1e-8, 1.0, &tt, 1, nullptr, 3, nullptr, nullptr, nullptr) != 0,
         "hb_adam_step info count without info");
  EXPECT(hb_gauss_ll_f32(fp, fp, nullptr, fp, 8, fp, fp, fp, fp, nullptr, 0, nullptr) != 0, "hb_gauss_ll no workspace");
  EXPECT(hb_gauss_ll_fold_f32(nullptr, 4, fp, fp, fp, nullptr) != 0, "hb_gauss_ll_fold NULL partials");
  EXPECT(hb_matmul_colsum_f32(fp, fp, fp, nullptr, 4, 4, 64, 4, 4, 4, fp, 1024, nullptr) != 0, "hb_matmul_colsum NULL colsum");
  EXPECT(std::strlen(hb_last_error_string()) > 0, "hb_last_error_string after a failure");
  std::printf(fails ? "host sanitizer driver: %d FAILED\n" : "host sanitizer driver: all checks passed (%d failures)\n", fails);
  return fails ? 1 : 0;
}
