// Descriptor of a fused elementwise program (hb_ewise_prog_*): shared by the interpreter kernels (elementwise.hip) and the
// run-time code generator (jit.hip).
#ifndef HB_EW_PROG_CUH
#define HB_EW_PROG_CUH
#define HB_PROG_MAX_INSTR 48
#define HB_PROG_MAX_IN 12
#define HB_PROG_MAX_OUT 6
#define HB_PROG_MAX_DIMS 4
#define HB_PROG_MAX_REGS 40
#define HB_PROG_SUM 256  // out_regs[k] flag: sum-reduce (forces a single workgroup; space <= HB_PROG_SUM_MAX_N)
#define HB_PROG_SUM_MAX_N 65536

struct ProgArgs {
  int ninstr, nin, nout, ndim;
  long n;
  int shape[HB_PROG_MAX_DIMS];
  long istr[HB_PROG_MAX_IN][HB_PROG_MAX_DIMS];
  long ostr[HB_PROG_MAX_OUT][HB_PROG_MAX_DIMS];
  const void* in[HB_PROG_MAX_IN];
  void* out[HB_PROG_MAX_OUT];
  int out_reg[HB_PROG_MAX_OUT];     // register written to out[k]; + HB_PROG_SUM: its SUM over the whole space -> out[k][0]
  short code[HB_PROG_MAX_INSTR][5];  // op, dst, a, b, c
  double params[HB_PROG_MAX_INSTR][2];
};

#endif  // HB_EW_PROG_CUH
