// kk_dist.hip -- 1-D row-partitioned SpMV across the GPUs of one node, behind the C ABI.
//
// The reference has no distributed layer (upstream that is Tpetra's job).  Rank r owns the contiguous row slab
// [offsets[r], offsets[r+1]) of A (local row_map, GLOBAL column indices), the matching slab of y and the matching shard of
// x.  One exchange step per SpMV, no other data-path communication:
//   all-gather  every rank receives every shard (ncclAllGather in place when the shards are equal, grouped send/recv
//               otherwise): what BASELINE.json's north star names;
//   halo        a rank needs only the x entries its slab's columns reference: for the column RANGE [cmin, cmax] of the
//               slab it receives from each peer the piece of that range the peer owns -- grouped ncclSend / ncclRecv,
//               point to point over xGMI.  For a slab of a 3-D stencil that is one grid plane per neighbour (5.8 MB per
//               rank at 600^3) instead of 1.5 GB.  Default when it moves less than half of the all-gather.
//   halo (column set)  for slabs whose columns are scattered (graphs: the column range is everything) the range halo degenerates
//               to the all-gather.  The importer instead finds the SET of off-slab columns the slab references (a flag per
//               column, compacted), tells every owner once which of its entries it wants, and per SpMV every rank packs the
//               requested entries of its shard (one gather kernel), the packed pieces travel point to point, and a scatter
//               kernel drops them into the full-length x -- the matrix keeps its global column indices.  Chosen by `auto`
//               when it moves less than half of what the range halo moves.
//   all-gather, peer to peer  the same result as the all-gather without a ring: every rank maps every other rank's x buffer
//               (hipIpc handles exchanged once) and PULLS the other shards with world - 1 concurrent copies on their own
//               streams, between two 8-byte collectives of the transport that act as stream-ordered barriers (all shards in
//               place before anybody pulls; all pulls done before anybody overwrites its shard).  xGMI is point to point
//               (7 links per GPU): seven concurrent 1/8-size copies use all links at once, where a ring all-gather is bound by
//               one link (SURVEY section 5 / 8e asks for this fallback in case RCCL schedules a ring).
// x lives in a full-length buffer owned by the operator (kkamd_dist_spmv_x_local hands out the rank's own window of it, so a
// solver that keeps its x there never copies it).  Overlap (halo mode): the rows that reference only the rank's own x
// entries -- the longest contiguous run of them -- are the INTERIOR, with their own plan over a zero-copy row-range view
// of the slab (rebased row_map); per SpMV the exchange runs on the operator's communication stream while the interior SpMV
// runs on the caller's stream, and the boundary rows wait for the exchange event.
//
// Transport.  The collective calls go through a small table (kkamd_transport_t): by default RCCL, bound at run time from
// the librccl.so.1 already in the process (torch's, MPI's or ROCm's: no second copy, no link-time dependency); a host
// that communicates otherwise (MPI, or the gloo process group the CPU tests use) passes its own two functions.
#include "kk_spmv_plan.h"
#include <cstring>
#include <new>
#include <string>
#include <chrono>
#include <cstring>
#include <vector>
#ifndef KK_EMU
#include <dlfcn.h>
#endif

namespace kk {

// ---- RCCL, bound at run time -------------------------------------------------------------------------------------
struct Rccl {
  typedef struct { char internal[128]; } UniqueId;
  typedef void* Comm;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  Rccl() {
#ifndef KK_EMU
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
#define KK_SYM(f) f = reinterpret_cast<decltype(f)>(dlsym(h, "nccl" #f)); if (!f) return
    KK_SYM(GetUniqueId); KK_SYM(CommInitRank); KK_SYM(CommDestroy); KK_SYM(AllGather); KK_SYM(Send); KK_SYM(Recv);
    KK_SYM(GroupStart); KK_SYM(GroupEnd); KK_SYM(GetErrorString);
#undef KK_SYM
    ok = true;
#endif
  }
};
static Rccl& rccl() { static Rccl r; return r; }
constexpr int kNcclInt8 = 0;      // ncclInt8 (rccl.h): payloads travel as bytes
#define KK_NCCL(expr)                                                                                                  \
  do {                                                                                                                 \
    int e_ = (expr);                                                                                                   \
    if (e_ != 0) return kk::fail(KKAMD_ERR_HIP, "%s failed: %s", #expr, kk::rccl().GetErrorString ? kk::rccl().GetErrorString(e_) : "?"); \
  } while (0)

struct RcclCtx { Rccl::Comm comm = nullptr; int world = 0, rank = 0; };

static int rccl_all_gather(void* ctx, const void* d_send, void* d_recv, int64_t bytes_per_rank, kkamd_stream_t stream) {
  RcclCtx* c = static_cast<RcclCtx*>(ctx);
  KK_NCCL(rccl().AllGather(d_send, d_recv, (size_t)bytes_per_rank, kNcclInt8, c->comm, to_hip(stream)));
  return KKAMD_OK;
}
static int rccl_exchange(void* ctx, int nsend, const void* const* d_send, const int64_t* send_bytes, const int* send_peer, int nrecv,
                         void* const* d_recv, const int64_t* recv_bytes, const int* recv_peer, kkamd_stream_t stream) {
  RcclCtx* c = static_cast<RcclCtx*>(ctx);
  KK_NCCL(rccl().GroupStart());
  // once the group is open it is always closed: a thread that returns between GroupStart and GroupEnd leaves RCCL's
  // per-thread group depth at one and every later collective of the process (torch's too) queues behind it for ever
  int first = 0; const char* what = "";
  for (int i = 0; i < nsend && !first; ++i) { first = rccl().Send(d_send[i], (size_t)send_bytes[i], kNcclInt8, send_peer[i], c->comm, to_hip(stream)); what = "ncclSend"; }
  for (int i = 0; i < nrecv && !first; ++i) { first = rccl().Recv(d_recv[i], (size_t)recv_bytes[i], kNcclInt8, recv_peer[i], c->comm, to_hip(stream)); what = "ncclRecv"; }
  const int end = rccl().GroupEnd();
  if (first) return kk::fail(KKAMD_ERR_HIP, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(first) : "?");
  if (end) return kk::fail(KKAMD_ERR_HIP, "ncclGroupEnd failed: %s", rccl().GetErrorString ? rccl().GetErrorString(end) : "?");
  return KKAMD_OK;
}

// ---- analysis kernels --------------------------------------------------------------------------------------------
__global__ void minmax_entries_kernel(const int32_t* __restrict__ entries, int64_t nnz, int* __restrict__ out) {
  int lo = INT32_MAX, hi = -1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = entries[i];
    lo = c < lo ? c : lo; hi = c > hi ? c : hi;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { atomicMin(out, lo); atomicMax(out + 1, hi); }
}
// flag[r] = 1 when row r references a column outside [c0, c1) (it depends on the halo)
template <class OffT>
__global__ void halo_rows_kernel(int64_t nrows, const OffT* __restrict__ row_map, const int32_t* __restrict__ entries, int c0, int c1,
                                 unsigned char* __restrict__ flag) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  unsigned char f = 0;
  for (OffT j = row_map[r]; j < row_map[r + 1]; ++j) { const int c = entries[j]; f |= (c < c0 || c >= c1) ? 1 : 0; }
  flag[r] = f;
}
// flag[c] = 1 for every column c outside [c0, c1) that the slab references
__global__ void mark_offslab_cols_kernel(const int32_t* __restrict__ entries, int64_t nnz, int c0, int c1, unsigned char* __restrict__ flag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = entries[i];
    if (c < c0 || c >= c1) flag[c] = 1;
  }
}
// dst[i] = src[idx[i]] (pack: src = the rank's shard, idx = what the peers asked for) / dst[idx[i]] = src[i] (scatter into x)
template <class T> __global__ void gather_idx_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, T* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
template <class T> __global__ void scatter_idx_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, T* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[idx[i]] = src[i];
}
__global__ void rebase_idx_kernel(int32_t* __restrict__ idx, int64_t n, int base) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] -= base;
}
template <class OffT>
__global__ void rebase_kernel(const OffT* __restrict__ row_map, int64_t a, int64_t count, OffT* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= count) out[i] = row_map[a + i] - row_map[a];
}

}  // namespace kk

struct kkamd_dist_spmv {
  kkamd_crs_t A{};
  std::vector<int64_t> offsets;
  int world = 1, rank = 0, algorithm = 0, elem = 8;
  kkamd_transport_t tr{};
  kk::RcclCtx rccl_ctx;                      // when the built-in transport is used
  bool own_comm = false;
  int mode = 0;                              // 0 local, 1 halo (column range), 2 all-gather, 3 all-gather by peer-to-peer pulls, 4 halo (column set)
  bool equal = true;
  int64_t exchange_bytes = 0, interior_rows = 0;
  void* d_x_full = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  std::vector<const void*> send_ptr; std::vector<int64_t> send_bytes; std::vector<int> send_peer;
  std::vector<void*> recv_ptr; std::vector<int64_t> recv_bytes; std::vector<int> recv_peer;
  // column-set halo (mode 4): what the peers want of my shard (local indices, peer after peer) and what I want (global columns)
  int32_t *d_send_idx = nullptr, *d_need_col = nullptr;
  void *d_pack = nullptr, *d_unpack = nullptr;
  int64_t n_send_idx = 0, n_need = 0;
  // peer-to-peer all-gather (mode 3): every rank's x buffer mapped here, one copy stream + event per peer, two barrier tokens
  std::vector<void*> peer_x;
  std::vector<hipStream_t> p2p_stream; std::vector<hipEvent_t> p2p_event;
  hipEvent_t ev_b1 = nullptr;
  void* d_tok = nullptr;
  // all-gather (mode 2 / 3): the form in use -- 0 the transport's collective, 1 every shard to every peer point to point (the transport's
  // send / receive groups), 2 peer-to-peer pulls of mapped memory -- and what one exchange of each form took when the operator timed them
  // at its creation (exchange 5; microseconds, maximum over the ranks; -1: not available / not timed)
  int ag_form = 0;
  double ag_us[3] = {-1.0, -1.0, -1.0};
  bool ag_selected = false;
  struct Part { kkamd_crs_t A{}; kkamd_spmv_plan_t* plan = nullptr; void* d_rm = nullptr; int64_t row0 = 0; };
  std::vector<Part> parts;                   // whole slab, or interior first and then the boundary parts
};

namespace kk {

static void dist_free(kkamd_dist_spmv* op) {
  if (!op) return;
  for (auto& p : op->parts) { if (p.plan) kkamd_spmv_plan_destroy(p.plan); if (p.d_rm) (void)hipFree(p.d_rm); }
#ifndef KK_EMU
  for (size_t p = 0; p < op->peer_x.size(); ++p) if (op->peer_x[p] && (int)p != op->rank) (void)hipIpcCloseMemHandle(op->peer_x[p]);
#endif
  for (auto s_ : op->p2p_stream) if (s_) (void)hipStreamDestroy(s_);
  for (auto e_ : op->p2p_event) if (e_) (void)hipEventDestroy(e_);
  if (op->ev_b1) (void)hipEventDestroy(op->ev_b1);
  for (void* b : {(void*)op->d_send_idx, (void*)op->d_need_col, op->d_pack, op->d_unpack, op->d_tok}) if (b) (void)hipFree(b);
  if (op->d_x_full) (void)hipFree(op->d_x_full);
  if (op->ev_ready) (void)hipEventDestroy(op->ev_ready);
  if (op->ev_done) (void)hipEventDestroy(op->ev_done);
  if (op->comm_stream) (void)hipStreamDestroy(op->comm_stream);
  if (op->own_comm && op->rccl_ctx.comm && rccl().ok) (void)rccl().CommDestroy(op->rccl_ctx.comm);
  delete op;
}

// One all-gather of the x shards in the given form, queued on the communication stream cs (see kkamd_dist_spmv::ag_form).
static int dist_allgather(kkamd_dist_spmv* op, int form, const void* x_local, int64_t mrows, hipStream_t cs) {
  kkamd_stream_t kcs = reinterpret_cast<kkamd_stream_t>(cs);
  if (form == 2) {
    // barrier (every shard is in place), world - 1 concurrent pulls on their own streams, barrier (nobody still reads a shard)
    char* tok = (char*)op->d_tok;
    int rc = op->tr.all_gather(op->tr.ctx, tok, tok + 8, 8, kcs);
    if (rc) return rc;
    KK_HIP(hipEventRecord(op->ev_b1, cs));
    size_t q = 0;
    for (int p = 0; p < op->world; ++p) {
      if (p == op->rank) continue;
      const int64_t off = (int64_t)op->elem * op->offsets[p], len = (int64_t)op->elem * (op->offsets[p + 1] - op->offsets[p]);
      hipStream_t s2 = op->p2p_stream[q];
      KK_HIP(hipStreamWaitEvent(s2, op->ev_b1, 0));
      if (len > 0) KK_HIP(hipMemcpyAsync((char*)op->d_x_full + off, (const char*)op->peer_x[(size_t)p] + off, (size_t)len, hipMemcpyDeviceToDevice, s2));
      KK_HIP(hipEventRecord(op->p2p_event[q], s2));
      KK_HIP(hipStreamWaitEvent(cs, op->p2p_event[q], 0));
      ++q;
    }
    return op->tr.all_gather(op->tr.ctx, tok, tok + 8, 8, kcs);
  }
  if (form == 0 && op->equal) return op->tr.all_gather(op->tr.ctx, x_local, op->d_x_full, (int64_t)op->elem * mrows, kcs);
  return op->tr.exchange(op->tr.ctx, (int)op->send_peer.size(), op->send_ptr.data(), op->send_bytes.data(), op->send_peer.data(),
                         (int)op->recv_peer.size(), op->recv_ptr.data(), op->recv_bytes.data(), op->recv_peer.data(), kcs);
}

template <class OffT>
static int dist_add_part(kkamd_dist_spmv* op, int64_t a, int64_t b, const std::vector<OffT>& h_rm, hipStream_t st) {
  if (b <= a) return KKAMD_OK;
  kkamd_dist_spmv::Part p;
  const int64_t count = b - a;
  const char* ent = (const char*)op->A.d_entries; const char* val = (const char*)op->A.d_values;
  const int64_t p0 = (int64_t)h_rm[a], p1 = (int64_t)h_rm[b];
  const int vsz = op->A.value_type == KKAMD_F64 ? 8 : 4;
  p.row0 = a;
  if (a == 0 && b == op->A.num_rows) {
    p.A = op->A;
  } else {
    KK_HIP(hipMalloc(&p.d_rm, sizeof(OffT) * (size_t)(count + 1)));
    KK_LAUNCH((rebase_kernel<OffT>), (unsigned)ceil_div(count + 1, kBlock), kBlock, 0, st, (const OffT*)op->A.d_row_map, a, count, (OffT*)p.d_rm);
    p.A = kkamd_crs_t{count, op->A.num_cols, p1 - p0, p.d_rm, ent + 4 * p0, val + (int64_t)vsz * p0, op->A.offset_type, op->A.value_type};
  }
  op->parts.push_back(p);
  KK_HIP(hipStreamSynchronize(st));
  int rc = kkamd_spmv_plan_create(&op->parts.back().plan, &op->parts.back().A, op->algorithm, reinterpret_cast<kkamd_stream_t>(st));
  return rc;
}

template <class OffT>
static int dist_setup(kkamd_dist_spmv* op, int exchange, int overlap, hipStream_t st) {
  const int64_t me0 = op->offsets[op->rank], me1 = op->offsets[op->rank + 1], n = op->offsets[op->world];
  const int world = op->world, me = op->rank;
  op->equal = true;
  for (int r = 0; r < world; ++r) op->equal = op->equal && (op->offsets[r + 1] - op->offsets[r] == op->offsets[1] - op->offsets[0]);
  std::vector<OffT> h_rm((size_t)op->A.num_rows + 1);
  KK_HIP(hipMemcpyAsync(h_rm.data(), op->A.d_row_map, sizeof(OffT) * h_rm.size(), hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  // one rank: nothing to exchange -- unless the caller asked for the built-in RCCL transport with a forced exchange (id128 given,
  // world = 1): then the one-rank communicator runs the same calls an N-rank job makes (ncclAllGather in place; a group with no peers)
  if (world == 1 && !(op->own_comm && exchange != 0)) { op->mode = 0; return dist_add_part<OffT>(op, 0, op->A.num_rows, h_rm, st); }
  // column range of the slab, exchanged once: every rank learns what every rank needs
  DevBuf mm, all;
  KK_HIP(mm.alloc(2 * sizeof(int64_t))); KK_HIP(all.alloc(2 * sizeof(int64_t) * (size_t)world));
  int h_mm[2] = {INT32_MAX, -1};
  KK_HIP(hipMemcpyAsync(mm.p, h_mm, sizeof h_mm, hipMemcpyHostToDevice, st));
  int* d_mm = mm.as<int>();
  if (op->A.nnz > 0) {
    KK_LAUNCH(minmax_entries_kernel, 1024, kBlock, 0, st, (const int32_t*)op->A.d_entries, op->A.nnz, d_mm);
    KK_LAUNCH_CHECK();
  }
  KK_HIP(hipMemcpyAsync(h_mm, mm.p, sizeof h_mm, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  int64_t h_pair[2] = {h_mm[0], h_mm[1]};
  KK_HIP(hipMemcpyAsync(mm.p, h_pair, sizeof h_pair, hipMemcpyHostToDevice, st));
  int rc = op->tr.all_gather(op->tr.ctx, mm.p, all.p, 2 * sizeof(int64_t), reinterpret_cast<kkamd_stream_t>(st));
  if (rc) return rc;
  std::vector<int64_t> h_all(2 * (size_t)world);
  KK_HIP(hipMemcpyAsync(h_all.data(), all.p, sizeof(int64_t) * h_all.size(), hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  const int64_t cmin = h_pair[0], cmax = h_pair[1];
  char* xf = (char*)op->d_x_full;
  const int64_t es = op->elem;
  int64_t halo_bytes = 0;
  for (int p = 0; p < world; ++p) {
    if (p == me) continue;
    int64_t lo = cmin > op->offsets[p] ? cmin : op->offsets[p], hi = (cmax + 1 < op->offsets[p + 1]) ? cmax + 1 : op->offsets[p + 1];   // what I need from p
    if (hi > lo) { op->recv_ptr.push_back(xf + es * lo); op->recv_bytes.push_back(es * (hi - lo)); op->recv_peer.push_back(p); halo_bytes += es * (hi - lo); }
    const int64_t plo = h_all[2 * p], phi = h_all[2 * p + 1];
    lo = plo > me0 ? plo : me0; hi = (phi + 1 < me1) ? phi + 1 : me1;                                                         // what p needs from me
    if (hi > lo) { op->send_ptr.push_back(xf + es * lo); op->send_bytes.push_back(es * (hi - lo)); op->send_peer.push_back(p); }
  }
  const int64_t full_bytes = es * (n - (me1 - me0));
  // the SET of off-slab columns the slab references (needed for the set halo, and to choose it): a flag per column, compacted on the host
  std::vector<std::vector<int32_t>> need((size_t)world);               // global columns wanted from each peer, ascending
  int64_t set_elems = 0;
  if (exchange == 0 || exchange == 4) {
    DevBuf flag;
    KK_HIP(flag.alloc((size_t)n));
    KK_HIP(hipMemsetAsync(flag.p, 0, (size_t)n, st));
    if (op->A.nnz > 0) {
      unsigned char* d_colflag = flag.as<unsigned char>();     // (the emulator's launch captures its arguments by value)
      KK_LAUNCH(mark_offslab_cols_kernel, 1024, kBlock, 0, st, (const int32_t*)op->A.d_entries, op->A.nnz, (int)me0, (int)me1, d_colflag);
      KK_LAUNCH_CHECK();
    }
    std::vector<unsigned char> h_flag((size_t)n);
    KK_HIP(hipMemcpyAsync(h_flag.data(), flag.p, (size_t)n, hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    for (int p = 0; p < world; ++p)
      for (int64_t c = op->offsets[p]; c < op->offsets[p + 1]; ++c) if (h_flag[(size_t)c]) need[(size_t)p].push_back((int32_t)c);
    for (int p = 0; p < world; ++p) set_elems += (int64_t)need[(size_t)p].size();
  }
  // every rank must take the same decision: the largest fractions over the ranks decide
  double frac = full_bytes > 0 ? (double)halo_bytes / (double)full_bytes : 0.0;
  double sfrac = full_bytes > 0 ? (double)(es * set_elems) / (double)full_bytes : 0.0;
  {
    double h_frac[2] = {frac, sfrac};
    KK_HIP(hipMemcpyAsync(mm.p, h_frac, sizeof h_frac, hipMemcpyHostToDevice, st));
    if ((rc = op->tr.all_gather(op->tr.ctx, mm.p, all.p, 2 * sizeof(int64_t), reinterpret_cast<kkamd_stream_t>(st)))) return rc;
    std::vector<double> h_f(2 * (size_t)world);
    KK_HIP(hipMemcpyAsync(h_f.data(), all.p, sizeof(double) * h_f.size(), hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    for (int p = 0; p < world; ++p) { frac = h_f[2 * p] > frac ? h_f[2 * p] : frac; sfrac = h_f[2 * p + 1] > sfrac ? h_f[2 * p + 1] : sfrac; }
  }
  // auto: the range halo when it moves less than half of the all-gather (contiguous pieces straight into x, no pack / scatter
  // kernels); else the set halo when THAT moves less than half of the all-gather; else the all-gather
  int chosen = exchange;
  if (exchange == 0) chosen = frac < 0.5 ? 1 : (sfrac < 0.5 ? 4 : 5);     // (an all-gather picks its own form: exchange 5)
  if (chosen == 2 || chosen == 3 || chosen == 5) {
    op->exchange_bytes = full_bytes;
    op->send_ptr.clear(); op->send_bytes.clear(); op->send_peer.clear(); op->recv_ptr.clear(); op->recv_bytes.clear(); op->recv_peer.clear();
    // every shard to every peer, point to point (what unequal shards always use, and one of the forms exchange 5 times)
    for (int p = 0; p < world; ++p) {
      if (p == me) continue;
      op->send_ptr.push_back(xf + es * me0); op->send_bytes.push_back(es * (me1 - me0)); op->send_peer.push_back(p);
      op->recv_ptr.push_back(xf + es * op->offsets[p]); op->recv_bytes.push_back(es * (op->offsets[p + 1] - op->offsets[p])); op->recv_peer.push_back(p);
    }
    op->ag_form = chosen == 3 ? 2 : (op->equal ? 0 : 1);
    op->mode = chosen == 3 ? 3 : 2;
    bool p2p_ok = false;
#ifndef KK_EMU
    if (chosen == 3 || (chosen == 5 && world > 1)) {
      // every rank's x buffer, mapped once: 64-byte handles through the transport.  Every step every rank takes is a step all ranks take (a rank
      // whose mapping fails still joins the collectives and says so in its flag): exchange 3 fails on every rank, exchange 5 goes on without this form
      hipIpcMemHandle_t mine; memset(&mine, 0, sizeof mine);
      bool ok = hipIpcGetMemHandle(&mine, op->d_x_full) == hipSuccess;
      if (!ok) (void)hipGetLastError();
      static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
      DevBuf hs, ha;
      KK_HIP(hs.alloc(64)); KK_HIP(ha.alloc(64 * (size_t)world));
      KK_HIP(hipMemcpyAsync(hs.p, &mine, 64, hipMemcpyHostToDevice, st));
      if ((rc = op->tr.all_gather(op->tr.ctx, hs.p, ha.p, 64, reinterpret_cast<kkamd_stream_t>(st)))) return rc;
      std::vector<hipIpcMemHandle_t> handles((size_t)world);
      KK_HIP(hipMemcpyAsync(handles.data(), ha.p, 64 * (size_t)world, hipMemcpyDeviceToHost, st));
      KK_HIP(hipStreamSynchronize(st));
      op->peer_x.assign((size_t)world, nullptr);
      op->peer_x[(size_t)me] = op->d_x_full;
      for (int p = 0; p < world && ok; ++p) {
        if (p == me) continue;
        if (hipIpcOpenMemHandle(&op->peer_x[(size_t)p], handles[(size_t)p], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); op->peer_x[(size_t)p] = nullptr; ok = false; break; }
        hipStream_t s2 = nullptr; hipEvent_t e2 = nullptr;
        if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
        op->p2p_stream.push_back(s2); op->p2p_event.push_back(e2);
      }
      if (ok && hipEventCreateWithFlags(&op->ev_b1, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ok = false; }
      KK_HIP(hipMalloc(&op->d_tok, 8 * (size_t)(world + 1)));
      KK_HIP(hipMemsetAsync(op->d_tok, 0, 8 * (size_t)(world + 1), st));
      // does EVERY rank have its mappings?  (8-byte flags through the same token buffer)
      unsigned long long flag = ok ? 1ull : 0ull;
      DevBuf fl; KK_HIP(fl.alloc(8 * (size_t)(world + 1)));
      KK_HIP(hipMemcpyAsync(fl.p, &flag, 8, hipMemcpyHostToDevice, st));
      if ((rc = op->tr.all_gather(op->tr.ctx, fl.p, (char*)fl.p + 8, 8, reinterpret_cast<kkamd_stream_t>(st)))) return rc;
      std::vector<unsigned long long> flags((size_t)world);
      KK_HIP(hipMemcpyAsync(flags.data(), (char*)fl.p + 8, 8 * (size_t)world, hipMemcpyDeviceToHost, st));
      KK_HIP(hipStreamSynchronize(st));
      p2p_ok = true;
      for (unsigned long long f : flags) p2p_ok = p2p_ok && f == 1ull;
      if (chosen == 3 && !p2p_ok)
        return fail(KKAMD_ERR_HIP, "kkamd_dist: mapping the ranks' x buffers into each other failed on some rank (HSA_ENABLE_IPC_MODE_LEGACY=0 set? peer access between the devices?)");
    }
#else
    if (chosen == 3) return fail(KKAMD_ERR_UNSUPPORTED, "kkamd_dist: the peer-to-peer all-gather maps device memory between processes (hipIpc): not under the emulator");
#endif
    if (chosen == 5 && world > 1) {
      // "validate RCCL's algorithm choice ... fall back to peer-to-peer".  One warm-up and two timed exchanges of every form the
      // transport and the runtime offer, on the communication stream, all ranks in step; the form with the smallest maximum over the ranks stays.
      void* x_local = (char*)op->d_x_full + es * me0;
      const int64_t mrows_ = me1 - me0;
      DevBuf tb; KK_HIP(tb.alloc(24 * (size_t)(world + 1)));
      double mine_us[3] = {-1.0, -1.0, -1.0};
      for (int form = 0; form < 3; ++form) {
        if ((form == 0 && !op->equal) || (form == 2 && !p2p_ok)) continue;
        for (int it = 0; it < 3; ++it) {
          KK_HIP(hipStreamSynchronize(op->comm_stream));
          const auto t0 = std::chrono::steady_clock::now();
          if ((rc = dist_allgather(op, form, x_local, mrows_, op->comm_stream))) return rc;
          KK_HIP(hipStreamSynchronize(op->comm_stream));
          const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
          if (it == 1 || (it == 2 && us < mine_us[form])) mine_us[form] = us;       // the better of the two timed ones
        }
      }
      KK_HIP(hipMemcpyAsync(tb.p, mine_us, 24, hipMemcpyHostToDevice, st));
      if ((rc = op->tr.all_gather(op->tr.ctx, tb.p, (char*)tb.p + 24, 24, reinterpret_cast<kkamd_stream_t>(st)))) return rc;
      std::vector<double> all((size_t)3 * world);
      KK_HIP(hipMemcpyAsync(all.data(), (char*)tb.p + 24, 24 * (size_t)world, hipMemcpyDeviceToHost, st));
      KK_HIP(hipStreamSynchronize(st));
      int best = -1;
      for (int form = 0; form < 3; ++form) {
        double mx = -1.0; bool have = true;
        for (int p = 0; p < world; ++p) { const double v = all[(size_t)3 * p + form]; if (v < 0) have = false; else if (v > mx) mx = v; }
        op->ag_us[form] = have ? mx : -1.0;
        if (have && (best < 0 || mx < op->ag_us[best])) best = form;
      }
      op->ag_form = best < 0 ? (op->equal ? 0 : 1) : best;
      op->mode = op->ag_form == 2 ? 3 : 2;
      op->ag_selected = true;
      if (g_verbose) printf("kkamd_dist rank %d: all-gather of %lld bytes per rank: collective %.0f us, send / receive %.0f us, peer-to-peer pulls %.0f us -> form %d\n",
                            me, (long long)(es * mrows_), op->ag_us[0], op->ag_us[1], op->ag_us[2], op->ag_form);
    }
    return dist_add_part<OffT>(op, 0, op->A.num_rows, h_rm, st);
  }
  if (chosen == 4) {
    // tell every owner which of its entries I want: counts first (so that everybody can size its buffers), then the index lists
    op->send_ptr.clear(); op->send_bytes.clear(); op->send_peer.clear(); op->recv_ptr.clear(); op->recv_bytes.clear(); op->recv_peer.clear();
    DevBuf cnt_s, cnt_a;
    KK_HIP(cnt_s.alloc(sizeof(int64_t) * (size_t)world)); KK_HIP(cnt_a.alloc(sizeof(int64_t) * (size_t)world * (size_t)world));
    std::vector<int64_t> h_cnt((size_t)world), h_cnt_all((size_t)world * (size_t)world);
    for (int p = 0; p < world; ++p) h_cnt[(size_t)p] = (int64_t)need[(size_t)p].size();
    KK_HIP(hipMemcpyAsync(cnt_s.p, h_cnt.data(), sizeof(int64_t) * (size_t)world, hipMemcpyHostToDevice, st));
    if ((rc = op->tr.all_gather(op->tr.ctx, cnt_s.p, cnt_a.p, (int64_t)sizeof(int64_t) * world, reinterpret_cast<kkamd_stream_t>(st)))) return rc;
    KK_HIP(hipMemcpyAsync(h_cnt_all.data(), cnt_a.p, sizeof(int64_t) * h_cnt_all.size(), hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    int64_t n_send = 0;                                        // what the peers want of my shard: row p of the count matrix, column me
    for (int p = 0; p < world; ++p) if (p != me) n_send += h_cnt_all[(size_t)p * world + me];
    op->n_need = set_elems; op->n_send_idx = n_send;
    KK_HIP(hipMalloc((void**)&op->d_need_col, sizeof(int32_t) * (size_t)(set_elems > 0 ? set_elems : 1)));
    KK_HIP(hipMalloc((void**)&op->d_send_idx, sizeof(int32_t) * (size_t)(n_send > 0 ? n_send : 1)));
    KK_HIP(hipMalloc(&op->d_pack, (size_t)es * (size_t)(n_send > 0 ? n_send : 1)));
    KK_HIP(hipMalloc(&op->d_unpack, (size_t)es * (size_t)(set_elems > 0 ? set_elems : 1)));
    {
      std::vector<int32_t> h_need; h_need.reserve((size_t)set_elems);
      for (int p = 0; p < world; ++p) h_need.insert(h_need.end(), need[(size_t)p].begin(), need[(size_t)p].end());
      if (set_elems) KK_HIP(hipMemcpyAsync(op->d_need_col, h_need.data(), sizeof(int32_t) * (size_t)set_elems, hipMemcpyHostToDevice, st));
      KK_HIP(hipStreamSynchronize(st));                        // h_need goes out of scope
    }
    // index lists: my wants go to the owners (as global columns; the owner rebases them), theirs come to me
    std::vector<const void*> sp; std::vector<int64_t> sb; std::vector<int> spr; std::vector<void*> rp; std::vector<int64_t> rb; std::vector<int> rpr;
    int64_t need_off = 0, send_off = 0;
    for (int p = 0; p < world; ++p) {
      const int64_t wn = (int64_t)need[(size_t)p].size(), ws = p == me ? 0 : h_cnt_all[(size_t)p * world + me];
      if (p != me && wn) {
        sp.push_back(op->d_need_col + need_off); sb.push_back(4 * wn); spr.push_back(p);
        op->recv_ptr.push_back((char*)op->d_unpack + es * need_off); op->recv_bytes.push_back(es * wn); op->recv_peer.push_back(p);       // per SpMV: values arrive here
      }
      if (p != me && ws) {
        rp.push_back(op->d_send_idx + send_off); rb.push_back(4 * ws); rpr.push_back(p);
        op->send_ptr.push_back((char*)op->d_pack + es * send_off); op->send_bytes.push_back(es * ws); op->send_peer.push_back(p);          // per SpMV: values leave from here
      }
      need_off += wn; send_off += ws;
    }
    if ((rc = op->tr.exchange(op->tr.ctx, (int)spr.size(), sp.data(), sb.data(), spr.data(), (int)rpr.size(), rp.data(), rb.data(), rpr.data(),
                              reinterpret_cast<kkamd_stream_t>(st)))) return rc;
    if (n_send) {                                              // global columns -> indices into my shard
      int32_t* d_sidx = op->d_send_idx;
      KK_LAUNCH(rebase_idx_kernel, (unsigned)ceil_div(n_send, kBlock), kBlock, 0, st, d_sidx, n_send, (int)me0);
      KK_LAUNCH_CHECK();
    }
    KK_HIP(hipStreamSynchronize(st));
    halo_bytes = es * set_elems;
  }
  op->mode = chosen; op->exchange_bytes = halo_bytes;           // 1 (column range) or 4 (column set)
  // interior = the longest contiguous run of rows that reference only this rank's own x entries
  int64_t r_lo = 0, r_hi = 0;
  const int64_t m = op->A.num_rows;
  if (overlap && m > 0 && op->A.nnz > 0) {
    DevBuf flag;
    KK_HIP(flag.alloc((size_t)m));
    unsigned char* d_flag = flag.as<unsigned char>();
    KK_LAUNCH((halo_rows_kernel<OffT>), (unsigned)ceil_div(m, kBlock), kBlock, 0, st, m, (const OffT*)op->A.d_row_map, (const int32_t*)op->A.d_entries,
              (int)me0, (int)me1, d_flag);
    KK_LAUNCH_CHECK();
    std::vector<unsigned char> h_flag((size_t)m);
    KK_HIP(hipMemcpyAsync(h_flag.data(), flag.p, (size_t)m, hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    int64_t best = 0, run0 = 0;
    for (int64_t r = 0; r <= m; ++r) {
      if (r == m || h_flag[(size_t)r]) { if (r - run0 > best) { best = r - run0; r_lo = run0; r_hi = r; } run0 = r + 1; }
    }
    // the planned kernel wants 16-byte aligned entries / values: the interior and the tail start on rows whose first entry
    // sits at a multiple of 4 (a misaligned view still works, through the no-analysis kernel)
    for (int k = 0; k < 64 && r_lo < r_hi && (h_rm[(size_t)r_lo] % 4) != 0; ++k) ++r_lo;
    for (int k = 0; k < 64 && r_hi > r_lo && (h_rm[(size_t)r_hi] % 4) != 0; ++k) --r_hi;
    if (r_hi - r_lo < m / 2) r_lo = r_hi = 0;                  // not worth splitting
  }
  if (r_hi > r_lo && !(r_lo == 0 && r_hi == m)) {
    op->interior_rows = r_hi - r_lo;
    if ((rc = dist_add_part<OffT>(op, r_lo, r_hi, h_rm, st))) return rc;
    if ((rc = dist_add_part<OffT>(op, 0, r_lo, h_rm, st))) return rc;
    return dist_add_part<OffT>(op, r_hi, m, h_rm, st);
  }
  op->interior_rows = (r_lo == 0 && r_hi == m) ? m : 0;        // nothing depends on the halo / no split
  return dist_add_part<OffT>(op, 0, m, h_rm, st);
}

}  // namespace kk

extern "C" {

int kkamd_dist_unique_id(void* id128) {
  if (!id128) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_unique_id: null buffer");
  if (!kk::rccl().ok) return kk::fail(KKAMD_ERR_UNSUPPORTED, "kkamd_dist: librccl.so.1 could not be loaded");
  kk::Rccl::UniqueId id;
  KK_NCCL(kk::rccl().GetUniqueId(&id));
  std::memcpy(id128, id.internal, 128);
  return KKAMD_OK;
}

// Preflight of the built-in transport on ONE rank: binds librccl, forms a one-rank communicator and runs every entry point the
// N-rank exchanges use -- ncclAllGather in place, and a group of ncclSend / ncclRecv to itself -- on `bytes` of device data that are
// checked afterwards.  A job can call it on every rank before the first operator is created: a binding or ABI problem then shows
// as an error code here, not as a hang in the first exchange.
int kkamd_dist_transport_selftest(int64_t bytes, kkamd_stream_t stream) {
  if (bytes < 1) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_transport_selftest: bytes must be positive");
#ifdef KK_EMU
  (void)stream;
  return kk::fail(KKAMD_ERR_UNSUPPORTED, "kkamd_dist_transport_selftest: RCCL is not available under the emulator");
#else
  if (!kk::rccl().ok) return kk::fail(KKAMD_ERR_UNSUPPORTED, "kkamd_dist: librccl.so.1 could not be loaded");
  hipStream_t st = kk::to_hip(stream);
  kk::Rccl::UniqueId id;
  KK_NCCL(kk::rccl().GetUniqueId(&id));
  kk::RcclCtx ctx; ctx.world = 1; ctx.rank = 0;
  KK_NCCL(kk::rccl().CommInitRank(&ctx.comm, 1, id, 0));
  struct Guard { kk::RcclCtx& c; ~Guard() { if (c.comm) (void)kk::rccl().CommDestroy(c.comm); } } guard{ctx};
  kk::DevBuf a, b;
  KK_HIP(a.alloc((size_t)bytes)); KK_HIP(b.alloc((size_t)bytes));
  std::vector<unsigned char> h((size_t)bytes), back((size_t)bytes);
  for (int64_t i = 0; i < bytes; ++i) h[(size_t)i] = (unsigned char)((i * 131 + 7) & 0xff);
  KK_HIP(hipMemcpyAsync(a.p, h.data(), (size_t)bytes, hipMemcpyHostToDevice, st));
  KK_HIP(hipMemsetAsync(b.p, 0, (size_t)bytes, st));
  // 1. the all-gather of one rank, in place (send == recv + rank * bytes), then out of place
  int rc = kk::rccl_all_gather(&ctx, a.p, a.p, bytes, reinterpret_cast<kkamd_stream_t>(st));
  if (rc) return rc;
  if ((rc = kk::rccl_all_gather(&ctx, a.p, b.p, bytes, reinterpret_cast<kkamd_stream_t>(st)))) return rc;
  KK_HIP(hipMemcpyAsync(back.data(), b.p, (size_t)bytes, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  if (std::memcmp(h.data(), back.data(), (size_t)bytes) != 0) return kk::fail(KKAMD_ERR_HIP, "kkamd_dist_transport_selftest: ncclAllGather returned other bytes than it was given");
  // 2. a group with one send and one receive, peer = this rank (what the halo exchanges do between neighbours)
  KK_HIP(hipMemsetAsync(b.p, 0, (size_t)bytes, st));
  const void* sp[1] = {a.p}; void* rp[1] = {b.p}; const int64_t nb[1] = {bytes}; const int peer[1] = {0};
  if ((rc = kk::rccl_exchange(&ctx, 1, sp, nb, peer, 1, rp, nb, peer, reinterpret_cast<kkamd_stream_t>(st)))) return rc;
  // 3. an empty group (a rank whose slab needs nothing from anybody)
  if ((rc = kk::rccl_exchange(&ctx, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, reinterpret_cast<kkamd_stream_t>(st)))) return rc;
  KK_HIP(hipMemcpyAsync(back.data(), b.p, (size_t)bytes, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  if (std::memcmp(h.data(), back.data(), (size_t)bytes) != 0) return kk::fail(KKAMD_ERR_HIP, "kkamd_dist_transport_selftest: the grouped ncclSend / ncclRecv to self returned other bytes than were sent");
  const int e = kk::rccl().CommDestroy(ctx.comm); ctx.comm = nullptr;
  if (e != 0) return kk::fail(KKAMD_ERR_HIP, "ncclCommDestroy failed: %s", kk::rccl().GetErrorString(e));
  return KKAMD_OK;
#endif
}

int kkamd_dist_spmv_create(kkamd_dist_spmv_t** out, const kkamd_crs_t* A_local, const int64_t* row_offsets, int world, int rank,
                           const void* id128, const kkamd_transport_t* transport, int algorithm, int exchange, int overlap,
                           int vector_type, kkamd_stream_t stream) {
  if (!out) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_create: null output pointer");
  *out = nullptr;
  int rc = kk::check_crs(A_local);
  if (rc) return rc;
  if (!row_offsets || world < 1 || rank < 0 || rank >= world) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_create: bad partition");
  if (exchange < 0 || exchange > 5)
    return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_create: exchange %d is not 0 (auto), 1 (halo, column range), 2 (all-gather, the transport's collective), 3 (all-gather by peer-to-peer pulls), 5 (all-gather, the fastest form timed at creation) or 4 (halo, column set)", exchange);
  if (vector_type != KKAMD_F32 && vector_type != KKAMD_F64) return kk::fail(KKAMD_ERR_UNSUPPORTED, "kkamd_dist_spmv_create: unsupported vector_type %d", vector_type);
  for (int r = 0; r < world; ++r) if (row_offsets[r + 1] < row_offsets[r]) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_create: row offsets must ascend");
  if (row_offsets[0] != 0 || A_local->num_rows != row_offsets[rank + 1] - row_offsets[rank])
    return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_create: the slab has %lld rows, the partition gives rank %d %lld", (long long)A_local->num_rows, rank,
                    (long long)(row_offsets[rank + 1] - row_offsets[rank]));
  if (A_local->num_cols != row_offsets[world]) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_create: column indices must be global (num_cols = total rows)");
  kkamd_dist_spmv* op = new (std::nothrow) kkamd_dist_spmv();
  if (!op) return kk::fail(KKAMD_ERR_ALLOC, "kkamd_dist_spmv_create: out of host memory");
  op->A = *A_local; op->offsets.assign(row_offsets, row_offsets + world + 1); op->world = world; op->rank = rank;
  op->algorithm = algorithm; op->elem = vector_type == KKAMD_F64 ? 8 : 4;
  hipStream_t st = kk::to_hip(stream);
  auto bail = [&](int code) { kk::dist_free(op); return code; };
  if (transport) {
    if (!transport->all_gather || !transport->exchange) return bail(kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_create: incomplete transport"));
    op->tr = *transport;
  } else if (world > 1 || (id128 && exchange != 0)) {
    if (!kk::rccl().ok) return bail(kk::fail(KKAMD_ERR_UNSUPPORTED, "kkamd_dist: librccl.so.1 could not be loaded"));
    if (!id128) return bail(kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_create: the built-in RCCL transport needs the unique id of kkamd_dist_unique_id"));
    kk::Rccl::UniqueId id; std::memcpy(id.internal, id128, 128);
    int e = kk::rccl().CommInitRank(&op->rccl_ctx.comm, world, id, rank);
    if (e != 0) return bail(kk::fail(KKAMD_ERR_HIP, "ncclCommInitRank failed: %s", kk::rccl().GetErrorString(e)));
    op->own_comm = true; op->rccl_ctx.world = world; op->rccl_ctx.rank = rank;
    op->tr = kkamd_transport_t{&op->rccl_ctx, kk::rccl_all_gather, kk::rccl_exchange};
  }
  const size_t xbytes = (size_t)op->elem * (size_t)(row_offsets[world] > 0 ? row_offsets[world] : 1);
  if (hipMalloc(&op->d_x_full, xbytes) != hipSuccess) return bail(kk::fail(KKAMD_ERR_ALLOC, "kkamd_dist_spmv_create: out of device memory for x (%zu bytes)", xbytes));
  // zero once: entries outside the slab's column range are never read, but must not be garbage NaNs for beta = 0 sanity checks
  if (hipMemsetAsync(op->d_x_full, 0, xbytes, st) != hipSuccess) return bail(kk::fail(KKAMD_ERR_HIP, "hipMemsetAsync failed"));
  if (world > 1 || op->own_comm) {
    if (hipStreamCreateWithFlags(&op->comm_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&op->ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&op->ev_done, hipEventDisableTiming) != hipSuccess)
      return bail(kk::fail(KKAMD_ERR_HIP, "kkamd_dist_spmv_create: could not create the communication stream"));
  }
  rc = A_local->offset_type == KKAMD_I64 ? kk::dist_setup<int64_t>(op, exchange, overlap, st) : kk::dist_setup<int32_t>(op, exchange, overlap, st);
  if (rc) return bail(rc);
  *out = op;
  return KKAMD_OK;
}

int kkamd_dist_spmv_destroy(kkamd_dist_spmv_t* op) { kk::dist_free(op); return KKAMD_OK; }

int kkamd_dist_spmv_x_local(kkamd_dist_spmv_t* op, void** d_x_local, void** d_x_full) {
  if (!op) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_x_local: null operator");
  if (d_x_local) *d_x_local = (char*)op->d_x_full + (int64_t)op->elem * op->offsets[op->rank];
  if (d_x_full) *d_x_full = op->d_x_full;
  return KKAMD_OK;
}

int kkamd_dist_spmv_query(const kkamd_dist_spmv_t* op, const char* key, int64_t* value) {
  if (!op || !key || !value) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_query: null argument");
  const std::string k(key);
  if (k == "exchange") *value = op->mode;                        // 0 local, 1 halo (range), 2 all-gather, 3 all-gather peer to peer, 4 halo (set)
  else if (k == "exchange_bytes") *value = op->exchange_bytes;   // bytes this rank receives per SpMV
  else if (k == "allgather_form") *value = (op->mode == 2 || op->mode == 3) ? (op->mode == 3 ? 2 : op->ag_form) : -1;   // 0 collective, 1 send / receive, 2 peer-to-peer pulls
  else if (k == "allgather_selected") *value = op->ag_selected ? 1 : 0;                                                // the operator timed the forms at its creation
  else if (k == "allgather_us_collective") *value = (int64_t)op->ag_us[0];
  else if (k == "allgather_us_sendrecv") *value = (int64_t)op->ag_us[1];
  else if (k == "allgather_us_p2p") *value = (int64_t)op->ag_us[2];
  else if (k == "interior_rows") *value = op->interior_rows;     // rows computed while the halo is in flight
  else if (k == "parts") *value = (int64_t)op->parts.size();
  else if (k == "sends") *value = (int64_t)op->send_peer.size();
  else if (k == "recvs") *value = (int64_t)op->recv_peer.size();
  else if (k == "part0_rows") *value = op->parts.empty() ? 0 : op->parts[0].A.num_rows;       // the interior view (or the whole slab) ...
  else if (k.rfind("part0_", 0) == 0) {                                                       // ... and what its plan's analysis produced: "part0_<plan query key>"
    if (op->parts.empty() || !op->parts[0].plan) { *value = 0; return KKAMD_OK; }
    return kkamd_spmv_plan_query(op->parts[0].plan, k.c_str() + 6, value);
  }
  else return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_query: unknown key '%s'", key);
  return KKAMD_OK;
}

// what: 0 = the whole step, 1 = the exchange only (measurement), 2 = the local SpMV only (x as it stands)
int kkamd_dist_spmv_apply(kkamd_dist_spmv_t* op, double alpha, const void* d_x_shard, double beta, void* d_y_shard, int what,
                          kkamd_stream_t stream) {
  if (!op) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_apply: null operator");
  if (what < 0 || what > 2) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_apply: what = %d is not 0 (exchange + SpMV), 1 (exchange) or 2 (SpMV)", what);
  if (what != 1 && !d_y_shard && op->offsets[op->rank + 1] > op->offsets[op->rank])
    return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spmv_apply: null y shard");
  hipStream_t st = kk::to_hip(stream);
  const int vt = op->elem == 8 ? KKAMD_F64 : KKAMD_F32;
  const int64_t me0 = op->offsets[op->rank], mrows = op->offsets[op->rank + 1] - me0;
  char* x_local = (char*)op->d_x_full + (int64_t)op->elem * me0;
  int rc;
  if (what != 2) {
    if (d_x_shard && d_x_shard != x_local && mrows > 0)          // a solver that keeps its x in x_local skips this copy
      KK_HIP(hipMemcpyAsync(x_local, d_x_shard, (size_t)op->elem * (size_t)mrows, hipMemcpyDeviceToDevice, st));
    if (op->mode != 0) {
      hipStream_t cs = op->comm_stream;
      KK_HIP(hipEventRecord(op->ev_ready, st));                  // x_local is in place, and earlier SpMVs are done with the halo
      KK_HIP(hipStreamWaitEvent(cs, op->ev_ready, 0));
      kkamd_stream_t kcs = reinterpret_cast<kkamd_stream_t>(cs);
      if (op->mode == 2 || op->mode == 3) {
        rc = kk::dist_allgather(op, op->mode == 3 ? 2 : op->ag_form, x_local, mrows, cs);
      } else if (op->mode == 4) {
        // pack what the peers asked for, the packed pieces travel point to point, scatter what arrived into the full-length x
        const int64_t ns = op->n_send_idx, nn = op->n_need;
        const int32_t* sidx = op->d_send_idx; const int32_t* ncol = op->d_need_col;
        void* pack = op->d_pack; void* unpack = op->d_unpack; void* xfull = op->d_x_full;
        if (ns) {
          if (op->elem == 8) KK_LAUNCH((kk::gather_idx_kernel<uint64_t>), (unsigned)kk::ceil_div(ns, kk::kBlock), kk::kBlock, 0, cs, (const uint64_t*)x_local, sidx, ns, (uint64_t*)pack);
          else KK_LAUNCH((kk::gather_idx_kernel<uint32_t>), (unsigned)kk::ceil_div(ns, kk::kBlock), kk::kBlock, 0, cs, (const uint32_t*)x_local, sidx, ns, (uint32_t*)pack);
        }
        rc = op->tr.exchange(op->tr.ctx, (int)op->send_peer.size(), op->send_ptr.data(), op->send_bytes.data(), op->send_peer.data(),
                             (int)op->recv_peer.size(), op->recv_ptr.data(), op->recv_bytes.data(), op->recv_peer.data(), kcs);
        if (!rc && nn) {
          if (op->elem == 8) KK_LAUNCH((kk::scatter_idx_kernel<uint64_t>), (unsigned)kk::ceil_div(nn, kk::kBlock), kk::kBlock, 0, cs, (const uint64_t*)unpack, ncol, nn, (uint64_t*)xfull);
          else KK_LAUNCH((kk::scatter_idx_kernel<uint32_t>), (unsigned)kk::ceil_div(nn, kk::kBlock), kk::kBlock, 0, cs, (const uint32_t*)unpack, ncol, nn, (uint32_t*)xfull);
        }
      } else rc = op->tr.exchange(op->tr.ctx, (int)op->send_peer.size(), op->send_ptr.data(), op->send_bytes.data(), op->send_peer.data(),
                                (int)op->recv_peer.size(), op->recv_ptr.data(), op->recv_bytes.data(), op->recv_peer.data(), kcs);
      if (rc) return rc;
      KK_HIP(hipEventRecord(op->ev_done, cs));
    }
  }
  const bool split = op->parts.size() > 1;
  for (size_t i = 0; i < op->parts.size(); ++i) {
    // the interior (part 0 of a split slab) needs no halo entry and runs while the exchange is in flight
    if (what != 2 && op->mode != 0 && (i == (split ? 1u : 0u))) KK_HIP(hipStreamWaitEvent(st, op->ev_done, 0));
    if (what == 1) continue;
    auto& p = op->parts[i];
    if (p.A.num_rows == 0) continue;
    if ((rc = kkamd_spmv(p.plan, &p.A, 'N', alpha, op->d_x_full, beta, (char*)d_y_shard + (int64_t)op->elem * p.row0, vt, stream))) return rc;
  }
  if (what == 1 && op->mode != 0) KK_HIP(hipStreamWaitEvent(st, op->ev_done, 0));     // exchange only: the caller's stream still waits for it
  return KKAMD_OK;
}

}  // extern "C"
