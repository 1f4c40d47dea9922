// grid_trace.cpp -- MEASUREMENT AID (tools/grid_model.py): runs the REAL 2-D block-cyclic scheduler
// (gpc_amd/csrc/grid_sched.hpp, the code libgpc_hip.so runs) over a GridOps / GridComm pair that executes nothing and
// writes down what the scheduler asked for: every kernel, copy, event record / wait and exchange of every rank, in host
// issue order, with the stream it went to and its size.  tools/grid_model.py replays these traces against measured
// single-GPU kernel times and a link bandwidth to predict the 1 / 2 / 4 / 8-GPU curve; tests/test_grid_model.py pins the
// trace to the scheduler's own gpc_grid_stats counts.  Plain C++, no HIP, no oracle; not part of the library.
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../gpc_amd/csrc/grid_sched.hpp"

namespace {
using namespace gpc::grid;

struct Sink {
  FILE* f = nullptr;
  int rank = 0;
  long idx = 0;
  void op(int st, const char* kind, const char* fmt = nullptr, ...) __attribute__((format(printf, 4, 5)))
  {
    fprintf(f, "{\"rank\":%d,\"i\":%ld,\"st\":%d,\"op\":\"%s\"", rank, idx++, st, kind);
    if(fmt) {
      va_list ap;
      va_start(ap, fmt);
      fputc(',', f);
      vfprintf(f, fmt, ap);
      va_end(ap);
    }
    fputs("}\n", f);
  }
};

// addresses nobody dereferences: the scheduler only does arithmetic on them
struct TraceOps : GridOps {
  Sink* s;
  uintptr_t next = (uintptr_t)1 << 40;
  long next_event = 1;
  double pending_flops = 0.0;
  explicit TraceOps(Sink* sink) : s(sink) {}
  int alloc(void** p, size_t bytes) override
  {
    *p = (void*)next;
    next += (bytes + 4095) & ~(size_t)4095;
    return GPC_OK;
  }
  int release(void*) override { return GPC_OK; }
  int upload(void*, const void*, size_t bytes) override
  {
    s->op(-1, "upload", "\"bytes\":%zu", bytes);
    return GPC_OK;
  }
  int download(void* dst, const void*, size_t bytes, int st) override
  {
    memset(dst, 0, bytes);
    s->op(st, "download", "\"bytes\":%zu", bytes);
    return GPC_OK;
  }
  int zero(void*, size_t bytes, int st) override
  {
    s->op(st, "zero", "\"bytes\":%zu", bytes);
    return GPC_OK;
  }
  int zero2d(double*, int64_t, int64_t m, int64_t n, int st) override
  {
    s->op(st, "zero", "\"bytes\":%lld", (long long)(8 * m * n));
    return GPC_OK;
  }
  int copy(void*, const void*, size_t bytes, int st) override
  {
    s->op(st, "copy", "\"bytes\":%zu", bytes);
    return GPC_OK;
  }
  void* event_create() override { return (void*)(uintptr_t)(next_event++); }
  void event_destroy(void*) override {}
  int record(void* ev, int st) override
  {
    s->op(st, "record", "\"ev\":%ld", (long)(uintptr_t)ev);
    return GPC_OK;
  }
  int wait(int st, void* ev) override
  {
    s->op(st, "wait", "\"ev\":%ld", (long)(uintptr_t)ev);
    return GPC_OK;
  }
  int sync(int st) override
  {
    s->op(st, "sync");
    return GPC_OK;
  }
  void* native_stream(int) override { return nullptr; }
  int gather_rows(const double*, int64_t, int64_t D, int64_t, int64_t, int64_t, int64_t ntiles, int64_t nb, double*, int64_t,
                  int st) override
  {
    s->op(st, "copy", "\"bytes\":%lld", (long long)(8 * ntiles * nb * D));
    return GPC_OK;
  }
  int gram_cross(const gpc_kspec*, const double*, int64_t Na, int64_t, const double*, int64_t Nb, int64_t, int64_t D, double*,
                 int64_t, int st) override
  {
    s->op(st, "gram", "\"rows\":%lld,\"cols\":%lld,\"D\":%lld", (long long)Na, (long long)Nb, (long long)D);
    return GPC_OK;
  }
  int gram_diag(const gpc_kspec*, const double*, int64_t N, int64_t, int64_t, double, double*, int st) override
  {
    s->op(st, "small", "\"bytes\":%lld", (long long)(8 * N));
    return GPC_OK;
  }
  int sum_host(const double*, int64_t n, double* out, int st) override
  {
    *out = (double)n;
    s->op(st, "download", "\"bytes\":8");
    return GPC_OK;
  }
  int fix_diag_pad(double*, const Layout& L, const double*, int st) override
  {
    s->op(st, "small", "\"bytes\":%lld", (long long)(8 * L.Lr * L.nb));
    return GPC_OK;
  }
  int put_rhs_rows(double*, int64_t, const double*, int64_t, int64_t d, const Layout& L, int st) override
  {
    s->op(st, "small", "\"bytes\":%lld", (long long)(8 * d * L.nloc));
    return GPC_OK;
  }
  int potrf_tile(double*, int64_t, int64_t n, int64_t, int*, int st) override
  {
    s->op(st, "potrf_tile", "\"n\":%lld", (long long)n);
    return GPC_OK;
  }
  int potrf_panel(int64_t M, int64_t nb, double*, int64_t, int64_t, int*, int st) override
  {
    s->op(st, "potrf_panel", "\"rows\":%lld,\"n\":%lld", (long long)M, (long long)nb);
    return GPC_OK;
  }
  // (a tall share: tile-inverse form without staging; priced as the one-call panel of nb + M rows, the recorded cost of which
  //  was measured on exactly that path)
  int potrf_panel_rows(int64_t M, int64_t nb, double*, int64_t, double*, int64_t, int64_t, int*, int st) override
  {
    if(M < 12288 || nb < 512 || nb > 2048 || M % 2) return GPC_EUNSUPPORTED;
    s->op(st, "potrf_panel", "\"rows\":%lld,\"n\":%lld", (long long)(M + nb), (long long)nb);
    return GPC_OK;
  }
  int trsm_rlt(const double*, int64_t, int64_t n, double*, int64_t, int64_t M, int st) override
  {
    s->op(st, "trsm_rlt", "\"rows\":%lld,\"n\":%lld", (long long)M, (long long)n);
    return GPC_OK;
  }
  int copy2d(double*, int64_t, const double*, int64_t, int64_t m, int64_t n, int st) override
  {
    if(m > 0 && n > 0) s->op(st, "copy", "\"bytes\":%lld", (long long)(8 * m * n));
    return GPC_OK;
  }
  int pack_tiles(double*, const double*, int64_t, int64_t, int64_t, int64_t count, int64_t nb, int st) override
  {
    if(count > 0) s->op(st, "copy", "\"bytes\":%lld", (long long)(8 * count * nb * nb));
    return GPC_OK;
  }
  int copy_tiles(double*, int64_t, int64_t, const double*, int64_t, int64_t, int64_t count, int64_t nb, int64_t ncols, int st) override
  {
    if(count > 0 && ncols > 0) s->op(st, "copy", "\"bytes\":%lld", (long long)(8 * count * nb * ncols));
    return GPC_OK;
  }
  int covgrad_local(double*, const Layout& L, const double*, int64_t, int64_t, double* trace, int st) override
  {
    *trace = 0.0;
    s->op(st, "small", "\"bytes\":%lld", (long long)(16 * L.Lr * L.nb * L.nloc));
    s->op(st, "download", "\"bytes\":8");
    return GPC_OK;
  }
  void prof_update_begin(double flops, int) override { pending_flops = flops; }
  int update(const UpdateArgs& u, int st) override
  {
    if(u.role == 3) {   // an update of the distributed inverse: priced like a trailing update of the same extent, named apart
      s->op(st, "inv_update", "\"flops\":%.17g,\"rows\":%lld,\"cols\":%lld,\"k\":%lld", 2.0 * (double)u.M * (double)u.Ncols * (double)u.K,
            (long long)u.M, (long long)u.Ncols, (long long)u.K);
      return GPC_OK;
    }
    s->op(st, "update", "\"flops\":%.17g,\"rows\":%lld,\"cols\":%lld,\"k\":%lld", pending_flops, (long long)u.M,
          (long long)u.Ncols, (long long)u.K);
    return GPC_OK;
  }
  int diag_logsum(const double*, const Layout&, double* out, int st) override
  {
    *out = 0.0;
    s->op(st, "download", "\"bytes\":8");
    return GPC_OK;
  }
  int rows_sumsq(const double*, int64_t, int64_t nrows, int64_t ncols, double* out, int st) override
  {
    for(int64_t e = 0; e < nrows; e++) out[e] = 0.0;
    s->op(st, "small", "\"bytes\":%lld", (long long)(8 * nrows * ncols));
    s->op(st, "download", "\"bytes\":%lld", (long long)(8 * nrows));
    return GPC_OK;
  }
  int gemm(char, char, int64_t M, int64_t N, int64_t K, double, const double*, int64_t, const double*, int64_t, double, double*,
           int64_t, int st) override
  {
    s->op(st, "gemm", "\"flops\":%.17g", 2.0 * (double)M * (double)N * (double)K);
    return GPC_OK;
  }
  int trsm_llt(const double*, int64_t, int64_t n, double*, int64_t, int64_t nrhs, int st) override
  {
    s->op(st, "trsm_l", "\"n\":%lld,\"nrhs\":%lld", (long long)n, (long long)nrhs);
    return GPC_OK;
  }
  int trsm_lln(const double*, int64_t, int64_t n, double*, int64_t, int64_t nrhs, int st) override
  {
    s->op(st, "trsm_l", "\"n\":%lld,\"nrhs\":%lld", (long long)n, (long long)nrhs);
    return GPC_OK;
  }
  int set_identity(double*, int64_t, int64_t n, int st) override
  {
    s->op(st, "zero", "\"bytes\":%lld", (long long)(8 * n * n));
    return GPC_OK;
  }
  int kern_grad_block(const gpc_kspec* ks, const double*, int64_t Na, int64_t, const double*, int64_t Nb, int64_t, int64_t,
                      const double*, int64_t, double* g, int st) override
  {
    for(int p = 0; p < ks->offs[ks->n_terms]; p++) g[p] = 0.0;
    s->op(st, "kern_grad", "\"rows\":%lld,\"cols\":%lld", (long long)Na, (long long)Nb);
    return GPC_OK;
  }
  int add_transposed(double*, int64_t, const double*, int64_t, int64_t n, int64_t d, int st) override
  {
    s->op(st, "small", "\"bytes\":%lld", (long long)(8 * n * d));
    return GPC_OK;
  }
  int read_info(const int*, int* out, int st) override
  {
    *out = 0;
    s->op(st, "download", "\"bytes\":4");
    return GPC_OK;
  }
};

// exchanges: every rank writes its own view (axis group, its index in it, the root, the pieces); the replay matches the
// n-th exchange of a group across its members
struct TraceComm : GridComm {
  Sink* s;
  int pr, pc, r, c;
  long seq[3] = {0, 0, 0};
  TraceComm(Sink* sink, int pr_, int pc_, int r_, int c_) : s(sink), pr(pr_), pc(pc_), r(r_), c(c_) {}
  int group_size(int axis) const override { return axis == AX_ROW ? pc : (axis == AX_COL ? pr : pr * pc); }
  int group_id(int axis) const { return axis == AX_ROW ? r : (axis == AX_COL ? c : 0); }
  int me(int axis) const { return axis == AX_ROW ? c : (axis == AX_COL ? r : r * pc + c); }
  int bcast(void*, int64_t count, int root, int axis, GridOps*, int st) override
  {
    if(group_size(axis) == 1) return GPC_OK;
    s->op(st, "bcast", "\"axis\":%d,\"group\":%d,\"seq\":%ld,\"me\":%d,\"root\":%d,\"bytes\":%lld", axis, group_id(axis), seq[axis]++,
          me(axis), root, (long long)(8 * count));
    return GPC_OK;
  }
  int allgatherv(void*, const int64_t*, const int64_t* count, int axis, GridOps*, int st) override
  {
    const int n = group_size(axis);
    if(n == 1) return GPC_OK;
    std::string pieces = "[";
    for(int i = 0; i < n; i++) pieces += (i ? "," : "") + std::to_string((long long)(8 * count[i]));
    pieces += "]";
    s->op(st, "allgatherv", "\"axis\":%d,\"group\":%d,\"seq\":%ld,\"me\":%d,\"pieces\":%s", axis, group_id(axis), seq[axis]++, me(axis),
          pieces.c_str());
    return GPC_OK;
  }
  int allreduce_dev(double*, int64_t count, int axis, GridOps*, int st) override
  {
    if(group_size(axis) == 1) return GPC_OK;
    s->op(st, "allreduce", "\"axis\":%d,\"group\":%d,\"seq\":%ld,\"me\":%d,\"bytes\":%lld", axis, group_id(axis), seq[axis]++, me(axis),
          (long long)(8 * count));
    return GPC_OK;
  }
  int host_reduce(int axis, int n)
  {
    if(group_size(axis) == 1) return GPC_OK;
    s->op(ST_MAIN, "allreduce_host", "\"axis\":%d,\"group\":%d,\"seq\":%ld,\"me\":%d,\"bytes\":%d", axis, group_id(axis), seq[axis]++,
          me(axis), 8 * n);
    return GPC_OK;
  }
  int allreduce_host(double*, int n, int axis) override { return host_reduce(axis, n); }
  int allmin_host(int64_t*) override { return host_reduce(AX_WORLD, 1); }
  int barrier() override { return host_reduce(AX_WORLD, 1); }
};
}  // namespace

// what: 1 = update_k (Gram + factor + log-det), 2 = factor only, 3 = update_k + alpha + gradient.  The trace of every rank
// goes to `path` as JSON lines; stats[rank*8 .. +8) receives the scheduler's own GridStats (as gpc_grid_stats returns them).
extern "C" int gridtrace_run(int pr, int pc, long nb, long N, long D, long d, long Ns, int lookahead, int what, const char* path,
                             double* stats)
{
  FILE* f = fopen(path, "w");
  if(!f) return -1;
  gpc_kspec ks;
  memset(&ks, 0, sizeof(ks));
  ks.n_terms = 2;
  ks.types[0] = GPC_KERN_RBF;
  ks.types[1] = GPC_KERN_WHITE;
  ks.offs[0] = 0; ks.offs[1] = 2; ks.offs[2] = 3;
  ks.params[0] = 1.0; ks.params[1] = 1.0; ks.params[2] = 0.1;
  static double dummy[1] = {0.0};
  int rc = GPC_OK;
  for(int rank = 0; rank < pr * pc && rc == GPC_OK; rank++) {
    Sink sink;
    sink.f = f;
    sink.rank = rank;
    const int r = rank / pc, c = rank % pc;
    GridGp gp(std::unique_ptr<GridOps>(new TraceOps(&sink)), std::unique_ptr<GridComm>(new TraceComm(&sink, pr, pc, r, c)), pr, pc,
              r, c, nb);
    gp.lookahead = lookahead ? 1 : 0;
    if(lookahead == 2) gp.panel_first = false;   // the free-running order (as gpc_grid_set_lookahead(g, 2))
    if(lookahead == 3) gp.panel_first = true;
    rc = gp.set_problem(&ks, dummy, N, D, N, d > 0 ? dummy : nullptr, d, N, Ns > 0 ? dummy : nullptr, Ns, Ns > 0 ? Ns : 1);
    sink.op(-1, "begin");     // everything before this line is problem set-up, not part of a step
    gp.reset_stats();
    int info = 0;
    double ld = 0.0, jit = 0.0;
    if(rc == GPC_OK) rc = what == 2 ? (gp.fill(0.0), gp.factor(&info)) : gp.update_k(&ld, &jit, &info);
    if(rc == GPC_OK && what == 3) {
      std::vector<double> g(64, 0.0);
      rc = gp.alpha(nullptr, 0);
      if(rc == GPC_OK) rc = gp.gradient(g.data());
    }
    if(stats) {
      const GridStats& st = gp.stats();
      double* o = stats + 8 * rank;
      o[0] = st.bytes_recv[0]; o[1] = st.bytes_recv[1]; o[2] = st.bytes_recv[2]; o[3] = (double)st.collectives;
      o[4] = st.update_flops; o[5] = (double)st.update_launches; o[6] = st.update_bytes; o[7] = 0.0;
    }
  }
  fclose(f);
  return rc;
}
