// kk_spmv_mvnnz.hip -- rank-2 CSR SpMV (SpMV_MV) split by NONZEROS, for matrices whose rows differ wildly in length (power-law
// graphs: R-MAT scale 22 has rows of 0 and of 1e5 entries around an average of 15).
//
// Reference: sparse/impl/KokkosSparse_spmv_impl.hpp:634-1004 gives every row to a team of fixed size, so a hub row keeps its team
// busy while the others idle; the rank-1 merge path (sparse/impl/KokkosSparse_spmv_impl_merge.hpp:37-330) balances by nonzeros but
// has no rank-2 form and finishes cut rows with atomics.  The wave-private gather kernel of kk_spmv_mv.hip (16 rows per wave, the
// longest row of the 16 sets the wave's trip count, rows above a threshold handed to a workgroup each) ran R-MAT scale 22 x 16 in
// 1.28 + 1.83 ms.  Here:
//
//   * the nonzero stream is cut into CHUNKS of 128 entries, one per 8-lane group (eight chunks per wave).  Lane j of a group carries
//     right-hand sides 2 j and 2 j + 1: an X row is eight 16-byte loads, a wave instruction gathers eight X rows.  A group walks its
//     chunk 8 entries at a time: the (column, row, value) triples arrive with one coalesced load each and are laid down in wave-private
//     LDS, all 8 X gathers are issued, then the products are added up run by run of equal row index.
//   * the plan keeps the ROW INDEX of every nonzero (4 bytes per nonzero, built once from row_map): a group finds its row boundaries
//     without searching row_map, and every group does the same amount of work whatever the row lengths.
//   * a run that ends inside the chunk is a finished row: y = beta y + alpha sum, stored by the group.  The first run of a chunk when it
//     continues the previous chunk's row, and the last run when the next chunk continues it, go to a CARRY slot of the chunk (head /
//     tail, 16 doubles each); a second small kernel adds the pieces of every cut row in chunk order and stores the row.  No atomics,
//     no pre-scaling pass, the same summation order on every run (deterministic).
//   * rows without entries are listed by the plan and get beta y from a third, tiny kernel.
#include "kk_spmv_plan.h"
#include "kk_scan.h"
#include <new>
#include <climits>

namespace kk {
constexpr int kMv6E = 128;       // entries per chunk (per 8-lane group)
struct alignas(16) Mv6Ent { int col, row; double val; };
}  // namespace kk

struct kkamd_mv6_plan {
  int64_t nchunks = 0, n_empty = 0;
  int32_t* d_rowid = nullptr;      // [nnz]
  int32_t* d_empty = nullptr;      // [n_empty] rows without entries
  double* d_carry = nullptr;       // [nchunks][2][16]: head / tail partial sums of the chunk
  size_t bytes = 0;
};

namespace kk {

void mv6_plan_destroy(kkamd_mv6_plan* p) {
  if (!p) return;
  if (p->d_rowid) (void)hipFree(p->d_rowid);
  if (p->d_empty) (void)hipFree(p->d_empty);
  if (p->d_carry) (void)hipFree(p->d_carry);
  delete p;
}
int64_t mv6_plan_query(const kkamd_mv6_plan* p, int what) {
  if (!p) return 0;
  switch (what) {
    case 0: return p->nchunks;
    case 1: return p->n_empty;
    case 2: return (int64_t)p->bytes;
    default: return 0;
  }
}

// row index of every nonzero: the last row whose range starts at or before the entry and is not empty
template <class OffT>
__global__ __launch_bounds__(kBlock) void mv6_rowid_kernel(int64_t nrows, int64_t nnz, const OffT* __restrict__ row_map, int32_t* __restrict__ rowid) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nnz) return;
  int64_t lo = 0, hi = nrows;                                    // largest r with row_map[r] <= i (then row_map[r + 1] > i: the row holds i)
  while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)row_map[mid] <= i) lo = mid; else hi = mid; }
  rowid[i] = (int32_t)lo;
}
// rows without entries, in ascending order (their beta y stores then fall into neighbouring lines): flag, prefix sum, compaction
template <class OffT>
__global__ __launch_bounds__(kBlock) void mv6_empty_flag_kernel(int64_t nrows, const OffT* __restrict__ row_map, int64_t* __restrict__ pos) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < nrows) pos[r] = (row_map[r + 1] == row_map[r]) ? 1 : 0;
  else if (r == nrows) pos[r] = 0;
}
__global__ __launch_bounds__(kBlock) void mv6_empty_compact_kernel(int64_t nrows, const int64_t* __restrict__ pos, int32_t* __restrict__ list) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < nrows && pos[r + 1] != pos[r]) list[pos[r]] = (int32_t)r;
}
__global__ __launch_bounds__(kBlock) void mv6_empty_rows_kernel(int64_t n, const int32_t* __restrict__ list, double* __restrict__ Y, int64_t ys0, int64_t ys1,
                                                                double beta, int ncv, int colmajor) {
  if (colmajor) {                                                // column-major Y: a lane per listed row, blockIdx.y = the column (the list ascends:
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;          // neighbouring lanes, neighbouring rows of one column)
    if (idx >= n) return;
    double* yp = Y + (int64_t)list[idx] * ys0 + (int64_t)blockIdx.y * ys1;
    *yp = (beta == 0.0) ? 0.0 : beta * (*yp);
    return;
  }
  // 16 lanes (one right-hand side each) take 16 consecutive rows of the list
  const int64_t g = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 16;
  const int j = threadIdx.x & 15;
  if (j >= ncv) return;
  for (int64_t idx = g * 16; idx < n && idx < g * 16 + 16; ++idx) {
    double* yp = Y + (int64_t)list[idx] * ys0 + j * ys1;
    *yp = (beta == 0.0) ? 0.0 : beta * (*yp);
  }
}

// carry slots of chunk g: head at (2 g) * 16, tail at (2 g + 1) * 16.
// Eight lanes per chunk, two right-hand sides per lane: an X row is eight 16-byte loads, a wave instruction gathers eight X rows.  X must
// be row-major with an even leading dimension and 16-byte aligned (the caller packs anything else).  Y_VEC: Y rows are 16-byte aligned
// pairs as well.  FULL: all 16 columns of the block exist.
// The walk is a three-stage pipeline over rounds of 8 entries: while round r is added up out of registers and LDS, the X gathers of
// round r + 1 and the (column, row, value) triples of round r + 2 are in flight (two LDS buffers for the triples, two register sets for
// the X values).  In the FULL form every load of the loop is unconditional (indices clamped, results of the entries past the chunk's end
// never added), so that the waits the compiler places count loads instead of draining them (measured without the pipeline, one round
// at a time: 1.52 ms on R-MAT scale 22 x 16 with 16 lanes x 8 bytes, 1.75 ms with 8 lanes x 16 bytes).
template <class AT, bool Y_VEC, bool FULL>
__global__ __launch_bounds__(kBlock) void spmv_mv6_kernel(int64_t nnz, const int32_t* __restrict__ entries, const AT* __restrict__ values,
                                                          const int32_t* __restrict__ rowid, const double* __restrict__ X, int64_t ldx,
                                                          double* __restrict__ Y, int64_t ys0, int64_t ys1, double alpha, double beta, int ncv,
                                                          double* __restrict__ carry, int remap, int x24) {
  using XV = kk_f64x2;
  constexpr int GL = 8, NG = kWave / GL, R = kMv6E / GL;         // lanes per chunk, chunks per wave, rounds per chunk
  // a chunk's 8 triples sit 9 slots apart from the next chunk's: the eight groups of a wave then read from eight different
  // quads of banks (at 8 slots = 128 bytes apart every group's read hit the same four banks)
  constexpr int GS = GL + 1;
  __shared__ Mv6Ent s_ent_all[kBlock / kWave][2][NG * GS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & (GL - 1), grp = lane / GL;
  const int64_t wave = xcd_order(blockIdx.x, gridDim.x, remap) * (kBlock / kWave) + w;
  if (wave * NG * kMv6E >= nnz) return;                          // the whole wave leaves together
  const int64_t g = wave * NG + grp;
  const int64_t e0 = g * kMv6E < nnz ? g * kMv6E : nnz, e1 = e0 + kMv6E < nnz ? e0 + kMv6E : nnz;
  const int prev_row = (e0 > 0 && e0 < nnz) ? rowid[e0 - 1] : -1;
  const int next_row = (e1 < nnz) ? rowid[e1] : -1;
  const bool have = FULL || 2 * j < ncv, have2 = FULL || 2 * j + 1 < ncv;   // the lane's two columns exist in this block
  const double* __restrict__ xc = X + (have ? 2 * j : 0);
  // X rows by 24-bit multiply + one 64-bit add when the column count and the leading dimension allow it (the general 64-bit product is
  // six instructions, two of them quarter rate, per gather)
  const bool small = x24 != 0;
  const unsigned ldx24 = (unsigned)ldx;
  double* __restrict__ cg = carry + g * 32;
  int cur_row = -1;
  bool open_left = false;
  double acc0 = 0.0, acc1 = 0.0;
  auto flush = [&]() {                                           // the run of cur_row ends inside the chunk
    if (open_left) { cg[2 * j] = acc0; cg[2 * j + 1] = acc1; }   // ... but began in an earlier chunk: its head piece
    else if (have) {
      double* yp = Y + (int64_t)cur_row * ys0 + 2 * j * ys1;
      if (Y_VEC && have2) {
        XV out;
        if (beta == 0.0) { out[0] = alpha * acc0; out[1] = alpha * acc1; }
        else { const XV old = *reinterpret_cast<const XV*>(yp); out[0] = beta * old[0] + alpha * acc0; out[1] = beta * old[1] + alpha * acc1; }
        *reinterpret_cast<XV*>(yp) = out;
      } else {
        yp[0] = (beta == 0.0) ? alpha * acc0 : beta * yp[0] + alpha * acc0;
        if (have2) yp[ys1] = (beta == 0.0) ? alpha * acc1 : beta * yp[ys1] + alpha * acc1;
      }
    }
    open_left = false;
  };
  // stage 1: the lane's triple of round rd (index clamped into the arrays; what lies past the chunk is never added)
  int t_col = 0, t_row = -1; AT t_val = AT(0);
  auto fetch = [&](int rd) {
    int64_t idx = e0 + rd * GL + j;
    if (idx > nnz - 1) idx = nnz - 1;
    t_col = entries[idx]; t_row = rowid[idx]; t_val = values[idx];
  };
  // stage 2: the triples of round rd into their LDS buffer, the round's X gathers into xs
  auto stage = [&](int rd, XV (&xs)[GL]) {
    Mv6Ent* buf = s_ent_all[w][rd & 1];
    Mv6Ent me; me.col = t_col; me.row = t_row; me.val = (double)t_val;
    KK_WAVE_SYNC();
    buf[grp * GS + j] = me;
    KK_WAVE_SYNC();
    fetch(rd + 1);                                               // (past the last round: one clamped, unused triple -- no branch around a load)
    const int nq = (int)(e1 - (e0 + rd * GL) < GL ? (e1 - (e0 + rd * GL) > 0 ? e1 - (e0 + rd * GL) : 0) : GL);
    KK_UNROLL
    for (int q = 0; q < GL; ++q) {
      const int col = buf[grp * GS + q].col;
      const double* xp = small ? xc + KK_UMUL24((unsigned)col, ldx24) : xc + (int64_t)col * ldx;
      if (FULL) xs[q] = *reinterpret_cast<const XV*>(xp);
      else {
        xs[q] = XV{0.0, 0.0};
        if (q < nq && have) { if (have2) xs[q] = *reinterpret_cast<const XV*>(xp); else xs[q][0] = *xp; }   // the odd last column: nothing is read past it
      }
    }
  };
  // stage 3: round rd added up, run by run of equal row index
  auto consume = [&](int rd, const XV (&xs)[GL]) {
    const Mv6Ent* buf = s_ent_all[w][rd & 1];
    const int nq = (int)(e1 - (e0 + rd * GL) < GL ? (e1 - (e0 + rd * GL) > 0 ? e1 - (e0 + rd * GL) : 0) : GL);
    KK_UNROLL
    for (int q = 0; q < GL; ++q) {
      if (q < nq) {
        const Mv6Ent e = buf[grp * GS + q];
        if (e.row != cur_row) {
          if (cur_row >= 0) flush();
          else open_left = (e.row == prev_row);                  // the chunk's first entry
          cur_row = e.row; acc0 = 0.0; acc1 = 0.0;
        }
        acc0 += e.val * xs[q][0]; acc1 += e.val * xs[q][1];
      }
    }
  };
  XV xa[GL], xb[GL];
  fetch(0);
  stage(0, xa);
  for (int rd = 0; rd < R - 2; rd += 2) {                        // every stage of the loop body unconditional; the last two rounds peeled
    stage(rd + 1, xb);
    consume(rd, xa);
    stage(rd + 2, xa);
    consume(rd + 1, xb);
  }
  stage(R - 1, xb);
  consume(R - 2, xa);
  consume(R - 1, xb);
  if (cur_row >= 0) {
    if (cur_row == next_row) { cg[(open_left ? 0 : 16) + 2 * j] = acc0; cg[(open_left ? 0 : 16) + 2 * j + 1] = acc1; }   // the next chunk goes on with
    else flush();                                                // this row: a tail piece (or, when the whole chunk is the middle of one row, another head piece)
  }
}

// every row cut by a chunk boundary: the chunk that holds the row's first cut -- its last run goes on in the next chunk and is not itself
// a continuation -- adds the head pieces of the chunks the row reaches into (their number follows from row_map: no search), in order,
// and stores the row.  16 lanes per chunk, eight independent loads in flight on a long chain.
template <class OffT>
__global__ __launch_bounds__(kBlock) void mv6_fixup_kernel(int64_t nnz, int64_t nchunks, const OffT* __restrict__ row_map, const int32_t* __restrict__ rowid,
                                                           const double* __restrict__ carry, double* __restrict__ Y, int64_t ys0, int64_t ys1,
                                                           double alpha, double beta, int ncv) {
  const int64_t g = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 16;
  const int j = threadIdx.x & 15;
  if (g >= nchunks) return;
  const int64_t e0 = g * kMv6E, e1 = e0 + kMv6E;
  if (e1 >= nnz) return;                                         // the last chunk has no successor
  const int rl = rowid[e1 - 1];
  if (rl != rowid[e1]) return;                                   // its last run ends with the chunk
  if (e0 > 0 && rowid[e0] == rl && rowid[e0 - 1] == rl) return;  // all middle: the chunk where the row begins does the sum
  const int64_t c_last = ((int64_t)row_map[rl + 1] - 1) / kMv6E; // the chunk that holds the row's last entry
  double sum = carry[g * 32 + 16 + j];
  int64_t c = g + 1;
  for (; c + 8 <= c_last + 1; c += 8) {
    double t[8];
    KK_UNROLL
    for (int u = 0; u < 8; ++u) t[u] = carry[(c + u) * 32 + j];
    KK_UNROLL
    for (int u = 0; u < 8; ++u) sum += t[u];
  }
  for (; c <= c_last; ++c) sum += carry[c * 32 + j];
  if (j < ncv) {
    double* yp = Y + (int64_t)rl * ys0 + j * ys1;
    *yp = (beta == 0.0) ? alpha * sum : beta * (*yp) + alpha * sum;
  }
}

template <class OffT>
static int mv6_plan_build_t(kkamd_spmv_plan* plan, const kkamd_crs_t* A, hipStream_t st) {
  plan->mv6_tried = true;
  if (A->nnz == 0 || A->num_rows == 0) return KKAMD_OK;
  kkamd_mv6_plan* p = new (std::nothrow) kkamd_mv6_plan();
  if (!p) return fail(KKAMD_ERR_ALLOC, "kkamd_spmv_mv: out of host memory");
  struct Guard { kkamd_mv6_plan* p; ~Guard() { if (p) mv6_plan_destroy(p); } } guard{p};
  p->nchunks = ceil_div(A->nnz, (int64_t)kMv6E);
  if (hipMalloc((void**)&p->d_rowid, sizeof(int32_t) * (size_t)A->nnz) != hipSuccess ||
      hipMalloc((void**)&p->d_carry, sizeof(double) * 32 * (size_t)p->nchunks) != hipSuccess) {
    (void)hipGetLastError();
    return KKAMD_OK;                                             // no memory for the plan: the row-based gather kernel serves the matrix
  }
  KK_LAUNCH((mv6_rowid_kernel<OffT>), (unsigned)ceil_div(A->nnz, kBlock), kBlock, 0, st, A->num_rows, A->nnz, (const OffT*)A->d_row_map, p->d_rowid);
  KK_LAUNCH_CHECK();
  DevBuf posb;
  if (posb.alloc(sizeof(int64_t) * (size_t)(A->num_rows + 1)) != hipSuccess) { (void)hipGetLastError(); return KKAMD_OK; }
  int64_t* d_pos = posb.as<int64_t>();
  const unsigned grid = (unsigned)ceil_div(A->num_rows + 1, kBlock);
  KK_LAUNCH((mv6_empty_flag_kernel<OffT>), grid, kBlock, 0, st, A->num_rows, (const OffT*)A->d_row_map, d_pos);
  KK_LAUNCH_CHECK();
  int rc = exclusive_scan_inplace<int64_t>(d_pos, A->num_rows + 1, st);
  if (rc) return rc;
  int64_t h_n = 0;
  KK_HIP(hipMemcpyAsync(&h_n, d_pos + A->num_rows, sizeof h_n, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  if (h_n > 0) {
    if (hipMalloc((void**)&p->d_empty, sizeof(int32_t) * (size_t)h_n) != hipSuccess) { (void)hipGetLastError(); return KKAMD_OK; }
    KK_LAUNCH(mv6_empty_compact_kernel, grid, kBlock, 0, st, A->num_rows, (const int64_t*)d_pos, p->d_empty);
    KK_LAUNCH_CHECK();
  }
  KK_HIP(hipStreamSynchronize(st));
  p->n_empty = (int64_t)h_n;
  p->bytes = sizeof(int32_t) * (size_t)A->nnz + 256 * (size_t)p->nchunks + sizeof(int32_t) * (size_t)h_n;
  plan->mv6 = p;
  guard.p = nullptr;
  return KKAMD_OK;
}
int mv6_plan_build(kkamd_spmv_plan* plan, const kkamd_crs_t* A, hipStream_t st) {
  return A->offset_type == KKAMD_I64 ? mv6_plan_build_t<int64_t>(plan, A, st) : mv6_plan_build_t<int32_t>(plan, A, st);
}

// Y(i, j) at i * ys0 + j * ys1 := beta * Y(i, j) + Yp(i, j), Yp row-major with leading dimension ldp: the way back from the row-major
// scratch a column-major Y is computed in (32 x 32 tiles through LDS: reads walk a row of Yp, writes walk a column of Y)
__global__ __launch_bounds__(kBlock) void mv6_unpack_kernel(int64_t n, int64_t nvec, const double* __restrict__ Yp, int64_t ldp, double* __restrict__ Y,
                                                            int64_t ys0, int64_t ys1, double beta) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  const int64_t i0 = (int64_t)blockIdx.x * 32;
  for (int64_t j0 = 0; j0 < nvec; j0 += 32) {
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {                             // read: consecutive lanes walk j (contiguous in Yp)
      const int64_t i = i0 + q, j = j0 + tx;
      tile[q][tx] = (i < n && j < nvec) ? Yp[i * ldp + j] : 0.0;
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {                             // write: consecutive lanes walk i
      const int64_t i = i0 + tx, j = j0 + q;
      if (i < n && j < nvec) { double* yp = Y + i * ys0 + j * ys1; *yp = (beta == 0.0) ? tile[tx][q] : beta * (*yp) + tile[tx][q]; }
    }
  }
}

template <class OffT, class AT>
static int mv6_launch(const kkamd_spmv_plan* plan, const kkamd_crs_t* A, const double* X, int64_t ldx, double* Y, int64_t ys0, int64_t ys1,
                      int64_t nvec, double alpha, double beta, hipStream_t st) {
  const kkamd_mv6_plan* p = plan->mv6;
  const unsigned grid = (unsigned)ceil_div(p->nchunks, (int64_t)8 * (kBlock / kWave));
  const bool yv = ys1 == 1 && ys0 % 2 == 0 && (uintptr_t)Y % 16 == 0;
  const int x24 = (A->num_cols < (1 << 24) && ldx < (1 << 24) && (double)A->num_cols * (double)ldx < 4.0e9) ? 1 : 0;
  for (int64_t c0 = 0; c0 < nvec; c0 += 16) {
    const int ncv = (int)(nvec - c0 < 16 ? nvec - c0 : 16);
    const double* Xb = X + c0;
    double* Yb = Y + c0 * ys1;
#define KK_MV6(YV, FU)                                                                                                                          \
    KK_LAUNCH((spmv_mv6_kernel<AT, YV, FU>), grid, kBlock, 0, st, A->nnz, (const int32_t*)A->d_entries, (const AT*)A->d_values, (const int32_t*)p->d_rowid, \
              Xb, ldx, Yb, ys0, ys1, alpha, beta, ncv, p->d_carry, plan->tune.mv_remap, x24)
    if (ncv == 16) { if (yv) KK_MV6(true, true); else KK_MV6(false, true); }
    else           { if (yv) KK_MV6(true, false); else KK_MV6(false, false); }
#undef KK_MV6
    KK_LAUNCH_CHECK();
    KK_LAUNCH((mv6_fixup_kernel<OffT>), (unsigned)ceil_div(p->nchunks * 16, kBlock), kBlock, 0, st, A->nnz, p->nchunks, (const OffT*)A->d_row_map,
              (const int32_t*)p->d_rowid, (const double*)p->d_carry, Yb, ys0, ys1, alpha, beta, ncv);
    KK_LAUNCH_CHECK();
    if (p->n_empty > 0) {
      if (ys0 < ys1) KK_LAUNCH(mv6_empty_rows_kernel, dim3((unsigned)ceil_div(p->n_empty, kBlock), (unsigned)ncv), kBlock, 0, st, p->n_empty, (const int32_t*)p->d_empty, Yb, ys0, ys1, beta, ncv, 1);
      else KK_LAUNCH(mv6_empty_rows_kernel, (unsigned)ceil_div(ceil_div(p->n_empty, (int64_t)16) * 16, kBlock), kBlock, 0, st, p->n_empty, (const int32_t*)p->d_empty, Yb, ys0, ys1, beta, ncv, 0);
      KK_LAUNCH_CHECK();
    }
  }
  return KKAMD_OK;
}

// X: row-major (element (i, k) at i * ldx + k), ldx even, 16-byte aligned
int mv6_spmv(kkamd_spmv_plan* plan, const kkamd_crs_t* A, const double* X, int64_t ldx, double* Y, int64_t ys0, int64_t ys1,
             int64_t nvec, double alpha, double beta, hipStream_t st) {
  // Column-major Y: a finished row would be 16 eight-byte stores into 16 different lines, whenever its last entry happens to be reached
  // (R-MAT scale 22 x 16: 2.56 ms against 1.29 for row-major Y).  The product goes to a row-major scratch of the plan instead and a
  // tiled transpose applies beta and writes Y in whole lines.
  if (ys0 < ys1 && nvec >= 2) {
    const int64_t ldp = (nvec + 1) & ~(int64_t)1;
    const size_t need = (size_t)A->num_rows * (size_t)ldp * sizeof(double);
    if (plan->ypack_bytes < need) {
      if (plan->d_ypack) { KK_HIP(hipStreamSynchronize(st)); KK_HIP(hipFree(plan->d_ypack)); plan->d_ypack = nullptr; plan->ypack_bytes = 0; }
      if (hipMalloc(&plan->d_ypack, need) != hipSuccess) { (void)hipGetLastError(); plan->d_ypack = nullptr; }
      else plan->ypack_bytes = need;
    }
    if (plan->d_ypack) {
      double* Yp = (double*)plan->d_ypack;
      const int rc = mv6_spmv(plan, A, X, ldx, Yp, ldp, 1, nvec, alpha, 0.0, st);
      if (rc) return rc;
      KK_LAUNCH(mv6_unpack_kernel, (unsigned)ceil_div(A->num_rows, (int64_t)32), kBlock, 0, st, A->num_rows, nvec, (const double*)Yp, ldp, Y, ys0, ys1, beta);
      KK_LAUNCH_CHECK();
      return KKAMD_OK;
    }
  }
  const bool o64 = A->offset_type == KKAMD_I64;
  if (A->value_type == KKAMD_F64)
    return o64 ? mv6_launch<int64_t, double>(plan, A, X, ldx, Y, ys0, ys1, nvec, alpha, beta, st) : mv6_launch<int32_t, double>(plan, A, X, ldx, Y, ys0, ys1, nvec, alpha, beta, st);
  return o64 ? mv6_launch<int64_t, float>(plan, A, X, ldx, Y, ys0, ys1, nvec, alpha, beta, st) : mv6_launch<int32_t, float>(plan, A, X, ldx, Y, ys0, ys1, nvec, alpha, beta, st);
}

}  // namespace kk
