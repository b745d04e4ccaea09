// kk_spmv_mv4.h -- the plane-marching rank-2 kernel (spmv_mv4_kernel), its gather kernel for the rows outside the stencil and their launch.
// In a header since round 6: the 192 instantiations of the kernel (offsets x matrix values x stencil x beta x X layout x partial block, 20 KB of
// code each) were 3.9 of the 6.5 MB of the rank-2 code object, which the runtime loads whole at the first rank-2 call of a process (19 of
// the 27 ms of config 3's first call).  Every (offset type, value type) pair is now instantiated in a translation unit of its own
// and every stencil pattern (kk_spmv_mv4_*.hip: explicit instantiations of launch_mv4_stencil), i.e. in a code object of its own, loaded when first launched.
#pragma once
#include "kk_spmv_plan.h"
#include <climits>

namespace kk {

// Plane-marching rank-2 kernel (mv_kernel 0 = auto, or 4; analysed handles, fp64, right-hand sides in blocks of 16, any strides).
//
// What the two kernels above cannot do is keep X inside the CU across MANY rows: the gather kernel re-reads every X row
// through the texture path once per nonzero, the LDS-staged tiles re-fetch their window once per 32 rows.  On a matrix whose
// rows are a radius-1 stencil on an nx x ny x nz lattice -- found, not assumed: the strides S2 (= nx ny) and S1 (= nx) come
// out of the columns of a few rows, the longest sampled row gives the offset list, and a device pass checks EVERY row against
// it -- a workgroup takes a patch of RI x RJ lattice points and MARCHES along the far stride: the X rows of three consecutive
// planes of the patch (plus a one-point halo) sit in an LDS ring of four slabs and an X row crosses L2 -> LDS about 1.6
// times per product instead of once per tile that touches it.  Rows that conform -- interior rows, and boundary rows whose
// missing entries point outside the lattice -- read no column information at all (8 B per nonzero, the values, plus two
// words per row: where the values start, which entries exist); the others (wrap-around couplings, anything irregular: 0.3 %
// of C3) are listed by the analysis and done by a small gather kernel afterwards.
//   One plane of the patch = 64 row PAIRS (8 lanes per pair, two right-hand sides per lane; the pair = two lattice rows that
// are neighbours in j, so that the X rows a stencil group touches serve both: Mv4Groups).  One workgroup fits a CU (133 KB
// of LDS), so HBM latency cannot be hidden by other workgroups; the loop is software-pipelined TWO planes deep instead: while
// plane k is computed out of LDS, the values of planes k + 1 and k + 2, the X slabs of planes k + 2 and k + 3 and the row
// words of plane k + 3 are in flight to two register sets (about 108 KB per CU), and the older set is written to the other
// half of the value buffer and to the free ring slot just before the plane's single barrier.  (Measured on the way,
// DESIGN 4.2 / profiles/round2: global_load_lds for all three streams 5.9 ms -- LDS-DMA sustains about 12 B/clk/CU with
// 16-byte pieces, a quarter of that with 4-byte ones --; values one step ahead with a wait per step 4.35 ms; one plane deep
// 4.05 ms; one row per lane 2.97 ms; this form 2.70 ms on C3, 0.96 of the HBM rate a streaming kernel reaches with its mix.)
constexpr int kMv4Threads = 512, kMv4RI = 32, kMv4RJ = 4, kMv4MaxL = 28;
__host__ __device__ constexpr int mv4_pitch(int ne, int vbytes) {   // entries per row of the value ring: the 4 rows of a read group on 4 different 16-B slots
  return ((ne * vbytes) % 128 == 0) ? ne + 2 * (8 / vbytes) : ne;
}

struct Mv4Tab { int n; int e[kMv4MaxL]; };   // analysis: n offsets col - row / the lattice steps of the entries
// The stencil as the kernel walks it: entries that differ only in their step along the NEAR stride's lines (dj) form a group
// (dk, di); a lane computes two lattice rows that are neighbours in j, so the four X rows j - 1 .. j + 2 of a group serve both.
constexpr int kMv4MaxG = 9;
struct Mv4Groups {
  int n, ng;                // entries, groups
  int e[kMv4MaxG];          // (dk + 1) | di << 2
  int pres[kMv4MaxG];       // bit d: the entry with dj = d - 1 exists
  int perm[kMv4MaxL];       // entry q -> its position 3 g + (dj + 1) in a row of the value buffer
  int ent[kMv4MaxL];        // entry q -> (dk + 1) | (dj (RI + 2) + di) << 2: plane and position relative to the row's own point (rank-1 kernel)
};

}  // namespace kk

struct kkamd_mv4_plan {
  int nx = 0, ny = 0, nz = 0, kc = 0, kc1 = 0, kc1_planes = 0;   // kc1 (knob value seen) / kc1_planes / nchunk1: the k-chunks of the rank-1 kernel
  int64_t S1 = 0, S2 = 0, npi = 0, npj = 0, nchunk = 0, nchunk1 = 0, n_nc = 0;
  kk::Mv4Groups grp{};               // the stencil in the kernel's order
  void* d_arow = nullptr;            // [rows] offset type of the matrix: where the row's values start when it conforms to the stencil, else -1
  uint32_t* d_amask = nullptr;       // [rows] which entries of the stencil the row holds (all of them away from the lattice boundary)
  int32_t* d_nc = nullptr;           // [n_nc] the rows that do not
  size_t bytes = 0;
};

namespace kk {

// rows outside the stencil pattern: 16 lanes per row (one right-hand side each).  The lanes fetch 16 entries of the row at a
// time (one each), then every lane walks all 16: the X reads of a chunk are independent loads (one contiguous 128 B each)
template <class OffT, class AT>
__global__ __launch_bounds__(kBlock) void mv4_rows_kernel(int64_t n_list, const int32_t* __restrict__ list, const OffT* __restrict__ row_map,
                                                          const int32_t* __restrict__ entries, const AT* __restrict__ values,
                                                          const double* __restrict__ X, int64_t xs0, int64_t xs1, double* __restrict__ Y,
                                                          int64_t ys0, int64_t ys1, double alpha, double beta, int ncv) {
  int64_t idx = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 16;
  const int j  = threadIdx.x & 15;                     // the lane's entry of a chunk of the row, and its column of the block
  const int jc = j < ncv ? j : ncv - 1;                // a block of fewer than 16 columns: the spare lanes read a valid column and store nothing
  const bool live = idx < n_list;                      // no early return: the shuffles below want whole groups
  if (!live) idx = n_list - 1;
  const int64_t r = list[idx];
  const int64_t b = (int64_t)row_map[r], e = (int64_t)row_map[r + 1];
  double acc = 0.0;
  for (int64_t a = b; a < e; a += 16) {
    const bool in = a + j < e;
    const int32_t my_col = in ? entries[a + j] : 0;
    const double my_val  = in ? (double)values[a + j] : 0.0;
    KK_UNROLL
    for (int q = 0; q < 16; ++q) {
      const int32_t col = __shfl(my_col, q, 16);
      const double v    = __shfl(my_val, q, 16);
      acc += v * X[(int64_t)col * xs0 + jc * xs1];
    }
  }
  if (!live || j >= ncv) return;
  double* yp = Y + r * ys0 + j * ys1;
  *yp = (beta == 0.0) ? alpha * acc : beta * (*yp) + alpha * acc;
}

// PART: a block of ncv < 16 right-hand sides (the remainder of a width that is no multiple of 16, or a narrow multivector): lane c
// still carries the columns 2 c and 2 c + 1, columns past the block are clamped to its last one when X is read (the same cache
// lines again: no extra traffic) and masked when Y is read or written.  Row-major X with an even ncv keeps the 16-byte loads
// (XROW: pieces past the block re-read its last piece); anything else goes element by element.
template <class OffT, class AT, int NG, unsigned PRES, bool BETA0, int XM, bool PART = false>
__global__ __launch_bounds__(kMv4Threads, 1) void spmv_mv4_kernel(const OffT* __restrict__ arow, const uint32_t* __restrict__ amask, const AT* __restrict__ values, Mv4Groups G,
                                                                  const double* __restrict__ X0, int64_t xs0, int64_t xs1, double* __restrict__ Y0,
                                                                  int64_t ys0, int64_t ys1, double alpha, double beta, int y_vec_ok, int nx,
                                                                  int ny, int nz, int64_t S1, int64_t S2, int64_t npi, int64_t npj, int kc, int ncv_last, int ncb) {
  constexpr bool XROW = XM == 1;                       // row-major X, 16-byte pieces
  constexpr bool XT   = XM == 2;                       // column-major X (unit stride along the rows): pieces dealt out column-wise, slab rows swizzled
  constexpr int RI = kMv4RI, RJ = kMv4RJ, W = RI + 2, SLABR = (RJ + 2) * W, SLABB = SLABR * 128, NT = kMv4Threads;
  constexpr int ROWS = NT / 8;                         // row PAIRS per plane: a lane owns the lattice rows (i, j) and (i, j + 1)
  constexpr int NP = SLABR * 8, NXP = (NP + NT - 1) / NT;   // 16-byte pieces per slab, per thread
  constexpr int NPOS = 3 * NG + (NG & 1);              // positions of a row of the value buffer: 3 per group, even
  constexpr int LP = mv4_pitch(NPOS, (int)sizeof(AT)); // entries per row of the value buffer
  constexpr int AV = (3 * NG + 7) / 8;                 // values a lane carries for a row: entries c, c + 8, ...
  using XV = kk_f64x2;
  using AV2 = typename vec2<AT>::type;
  KK_DYN_SMEM(char, smem);                             // [X ring: 4 slabs][values: 2 buffers x 2 rows of a pair x ROWS x LP]
  __shared__ int perm_s[kMv4MaxL];
  char* ring = smem;
  AT* abuf   = reinterpret_cast<AT*>(smem + 4 * (size_t)SLABB);
  const int t = threadIdx.x, rs = t >> 3, line = rs / RI, ii = rs % RI, c = t & 7;
  // ncb blocks of 16 right-hand sides in one launch: the workgroups of one (patch, k-chunk) for the ncb blocks are neighbours in the launch
  // order, so they run side by side in one XCD and the matrix's values, which all of them read, cross HBM -> L2 once for the ncb of them
  const int64_t bb = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring patches of one k-chunk meet in one XCD's L2
  const int64_t b = ncb > 1 ? bb / ncb : bb;
  const int64_t cblk = ncb > 1 ? bb - b * ncb : 0;
  const double* __restrict__ X = X0 + cblk * 16 * xs1;
  double* __restrict__ Y = Y0 + cblk * 16 * ys1;
  // (PART with several blocks: the blocks before the last are full ones in the partial form -- a width that is no multiple of 16 in ONE
  // launch, so that the blocks' workgroups also share the X rows: with a row pitch of 192 or 320 bytes a block's 128 bytes of a row are
  // parts of two lines, and the other parts belong to the neighbour block)
  const int ncv = (PART && cblk != ncb - 1) ? 16 : ncv_last;
  const int64_t npatch = npi * npj;
  const int64_t ch = b / npatch, p = b % npatch;
  const int i0 = (int)(p % npi) * RI, j0 = (int)(p / npi) * RJ;
  const int kbeg = (int)ch * kc, kend = (kbeg + kc < nz) ? kbeg + kc : nz;
  const int njj = (j0 + RJ <= ny) ? RJ : ny - j0;
  // Addresses are a per-lane base (plane 0 of the lattice; computed once) plus a per-plane scalar offset: the 64-bit products
  // stay on the scalar unit, a load costs the vector unit one 64-bit add.  Lanes without a row, halo points outside the
  // lattice and planes past the end use a clamped, legal address: every load of the loop is UNCONDITIONAL, because the
  // compiler counts the vector-memory operations between a load and its use to place s_waitcnt vmcnt(N) and a load it may
  // have branched around counts as zero -- one conditional load younger than the awaited one turns the wait into vmcnt(0).
  // And nothing touches a loaded register before its real use (a select or a sign extension right after the load is a use:
  // the wait would sit there): row words, masks and values stay raw, validity is applied where they are consumed.
  bool lane_ok[2];
  const OffT* wbase[2]; const uint32_t* mbase[2]; double* ybase[2];
  KK_UNROLL
  for (int u = 0; u < 2; ++u) {
    const int jj = 2 * line + u;
    lane_ok[u] = jj < njj && i0 + ii < nx;
    const int64_t r0 = lane_ok[u] ? (int64_t)(j0 + jj) * S1 + i0 + ii : 0;
    wbase[u] = arow + r0; mbase[u] = amask + r0;
    ybase[u] = Y + r0 * ys0 + (2 * c) * ys1;
  }
  const bool ycol0 = !PART || 2 * c < ncv, ycol1 = !PART || 2 * c + 1 < ncv;      // which of the lane's two columns the block has
  const double* xbase[NXP];
  unsigned xsec_none = 0;
  bool x_in[NXP];                                      // the piece is a lattice point of the plane (else the halo holds 0)
  // Column-major X (XT): a piece is two 8-byte loads, and with eight consecutive lanes on the eight pieces of one X row a wave's
  // load touches 8 columns x 64 bytes.  There the pieces are dealt out the other way round -- 16 consecutive slab rows of one piece
  // to 16 consecutive lanes, a wave covers 4 pieces x 16 rows: whole 128-byte lines of 4 columns -- and the slab's rows are
  // swizzled in LDS (piece p of a row whose point is i sits at slot p ^ (i & 7)) so that these column-wise stores spread over all
  // banks; the compute phase reads a row's eight pieces as before, permuted inside the same 128 bytes.
  int x_dst[XT ? NXP : 1];                             // XT: byte offset of the piece inside its slab
  unsigned x_live = 0;                                 // XT: the thread has a piece in this round
  KK_UNROLL
  for (int it = 0; it < NXP; ++it) {
    int g = it * NT + t;
    int xr, part;
    if constexpr (!XT) { g = g < NP ? g : NP - 1; xr = g >> 3; part = g & 7; }
    else {
      // 64 consecutive lanes on 64 consecutive slab rows of one piece: a load is two runs (272 and 240 bytes: 34 points of a slab line, 30 of the
      // next).  (16 rows per piece and four pieces per load -- four runs of 128 bytes, each part of two lines -- measured 0.1 ms slower on C3.)
      part = g / SLABR; xr = g - part * SLABR;
      if (part > 7) { part = 7; xr = SLABR; }
      if (xr < SLABR) x_live |= 1u << it; else xr = SLABR - 1;
      x_dst[it] = (xr * 8 + (part ^ ((xr % W) & 7))) * 16;
    }
    const int jr = j0 - 1 + xr / W, ir = i0 - 1 + xr % W;
    x_in[it] = jr >= 0 && jr < ny && ir >= 0 && ir < nx;
    const int jq = jr < 0 ? 0 : (jr > ny - 1 ? ny - 1 : jr), iq = ir < 0 ? 0 : (ir > nx - 1 ? nx - 1 : ir);
    const int col0 = !PART ? 2 * part : (XROW ? (2 * part + 1 < ncv ? 2 * part : ncv - 2) : (2 * part < ncv ? 2 * part : ncv - 1));
    xbase[it] = X + ((int64_t)jq * S1 + iq) * xs0 + col0 * xs1;
    if (PART && !XROW && 2 * part + 1 >= ncv) xsec_none |= 1u << it;       // the piece's second column is past the block: it reads the first again
  }
  auto plane_clamped = [&](int kp) -> int64_t { return kp < 0 ? 0 : (kp > nz - 1 ? nz - 1 : kp); };   // scalar
  auto conforms = [&](int k, int u, OffT w) { return lane_ok[u] && k < kend && w >= 0; };
  auto load_words = [&](int kp, OffT (&w)[2], uint32_t (&mk)[2]) {          // raw: where the rows' values start (or -1), which entries they hold
    const int64_t off = plane_clamped(kp) * S2;
    KK_UNROLL
    for (int u = 0; u < 2; ++u) { w[u] = wbase[u][off]; mk[u] = mbase[u][off]; }
  };
  auto load_slab = [&](int kp, XV (&rx)[NXP]) {        // plane kp of the patch (with its halo): this thread's pieces
    const int64_t off = plane_clamped(kp) * S2 * xs0;
    KK_UNROLL
    for (int it = 0; it < NXP; ++it) {
      if constexpr (XROW) rx[it] = *reinterpret_cast<const XV*>(xbase[it] + off);        // row-major X: the piece is 16 contiguous bytes
      else if constexpr (PART) { rx[it][0] = xbase[it][off]; rx[it][1] = xbase[it][off + (((xsec_none >> it) & 1u) ? 0 : xs1)]; }
      else { rx[it][0] = xbase[it][off]; rx[it][1] = xbase[it][off + xs1]; }               // any strides (LayoutLeft: a wave's 8 pieces of a column are 64 contiguous bytes)
    }
  };
  // halo points outside the lattice hold 0 (boundary rows meet them with the value 0: no 0 * Inf): zeros go in when the slot
  // is free, the loaded pieces -- lattice points only, a predicated store, no select on a loaded register -- at the end
  auto zero_halo = [&](int kp) {
    if (kp > kend) return;
    char* dst = ring + (size_t)(kp & 3) * SLABB;
    const bool plane_in = kp >= 0 && kp < nz;
    const XV zero = {0.0, 0.0};
    KK_UNROLL
    for (int it = 0; it < NXP; ++it) {
      if constexpr (!XT) {
        const int g = it * NT + t;
        if (g < NP && !(plane_in && x_in[it])) *reinterpret_cast<XV*>(dst + (size_t)g * 16) = zero;
      } else {
        if (((x_live >> it) & 1u) && !(plane_in && x_in[it])) *reinterpret_cast<XV*>(dst + x_dst[it]) = zero;
      }
    }
  };
  auto store_slab = [&](int kp, const XV (&rx)[NXP]) {
    if (kp > kend || kp < 0 || kp >= nz) return;       // workgroup-uniform; no row of this chunk references a plane past kend
    char* dst = ring + (size_t)(kp & 3) * SLABB;
    KK_UNROLL
    for (int it = 0; it < NXP; ++it) {
      if constexpr (!XT) {
        const int g = it * NT + t;
        if (g < NP && x_in[it]) *reinterpret_cast<XV*>(dst + (size_t)g * 16) = rx[it];
      } else {
        if (((x_live >> it) & 1u) && x_in[it]) *reinterpret_cast<XV*>(dst + x_dst[it]) = rx[it];
      }
    }
  };
  const uint32_t full = G.n >= 32 ? 0xffffffffu : ((1u << G.n) - 1u);
  KK_UNROLL
  for (int q = 0; q < kMv4MaxL; ++q) if (t == q) perm_s[q] = G.perm[q];
  __syncthreads();
  int qoff[AV], qpos[AV];                              // entry c + 8 m of a full row, clamped into the row; its position in the value buffer
  KK_UNROLL
  for (int m = 0; m < AV; ++m) { qoff[m] = (c + 8 * m < G.n) ? c + 8 * m : G.n - 1; qpos[m] = perm_s[qoff[m]]; }
  auto load_values = [&](int k, const OffT (&w)[2], const uint32_t (&mk)[2], AT (&ra)[2][AV]) {   // raw; rows that do not conform load the head of the array
    KK_UNROLL
    for (int u = 0; u < 2; ++u) {
      const bool cf = conforms(k, u, w[u]);
      const AT* vb = values + (cf ? (int64_t)w[u] : 0);
      const bool part = cf && mk[u] != full;           // a boundary row: stencil entry q sits at the packed position = entries held before it
      if (!__any(part)) {
        KK_UNROLL
        for (int m = 0; m < AV; ++m) ra[u][m] = vb[qoff[m]];
      } else {
        const uint32_t mask = cf ? mk[u] : 1u;
        KK_UNROLL
        for (int m = 0; m < AV; ++m) {
          const int q = c + 8 * m;
          ra[u][m] = vb[((mask >> q) & 1u) ? __popc(mask & ((1u << q) - 1u)) : 0];
        }
      }
    }
  };
  auto store_values = [&](int buf, const uint32_t (&mk)[2], const AT (&ra)[2][AV]) {      // entries the row lacks carry 0
    KK_UNROLL
    for (int u = 0; u < 2; ++u) {
      AT* dst = abuf + ((size_t)(buf * 2 + u) * ROWS + rs) * LP;
      KK_UNROLL
      for (int m = 0; m < AV; ++m) { const int q = c + 8 * m; if (q < G.n) dst[qpos[m]] = ((mk[u] >> q) & 1u) ? ra[u][m] : AT(0); }
    }
  };
  auto y_ptr = [&](int k, int u) -> double* { return ybase[u] + (int64_t)k * S2 * ys0; };
  auto load_yold = [&](int k, const OffT (&w)[2], XV (&yo)[2]) {
    KK_UNROLL
    for (int u = 0; u < 2; ++u) {
      if (conforms(k, u, w[u])) {
        const double* yp = y_ptr(k, u);
        if constexpr (PART) { if (ycol1 && y_vec_ok) yo[u] = *reinterpret_cast<const XV*>(yp); else { if (ycol0) yo[u][0] = yp[0]; if (ycol1) yo[u][1] = yp[ys1]; } }
        else if (y_vec_ok) yo[u] = *reinterpret_cast<const XV*>(yp); else { yo[u][0] = yp[0]; yo[u][1] = yp[ys1]; }
      }
    }
  };

  // Row words (and entry masks) of plane p are loaded at the top of plane p - 3 into the stage of that plane's parity, read
  // from there for the value addresses at the top of plane p - 2, moved to `nxt` at the end of plane p - 2 (nothing moves a
  // register in the plane that loads it: the move would be a wait inside the youngest batch), to `cur` a plane later.
  OffT w_cur[2], w_nxt[2], w_stage[2][2];
  uint32_t m_cur[2], m_nxt[2], m_stage[2][2];
  AT ra[2][2][AV];                                     // two sets in flight: set P is loaded in planes of parity P ...
  XV rx[2][NXP];                                       // ... and written to LDS at the end of the next plane
  // prologue: the row words of the first three planes, the values of the first, three planes of X; then the first set in flight
  load_words(kbeg, w_cur, m_cur); load_words(kbeg + 1, w_nxt, m_nxt); load_words(kbeg + 2, w_stage[1], m_stage[1]);
  load_values(kbeg, w_cur, m_cur, ra[0]);
  {
    XV rt[NXP];                                        // three slabs requested before the first is awaited
    load_slab(kbeg - 1, rx[0]); load_slab(kbeg, rx[1]); load_slab(kbeg + 1, rt);
    zero_halo(kbeg - 1); zero_halo(kbeg); zero_halo(kbeg + 1); zero_halo(kbeg + 2);
    store_slab(kbeg - 1, rx[0]); store_slab(kbeg, rx[1]); store_slab(kbeg + 1, rt);
  }
  store_values(0, m_cur, ra[0]);
  load_slab(kbeg + 2, rx[1]);
  load_values(kbeg + 1, w_nxt, m_nxt, ra[1]);
  __syncthreads();
  for (int kk = kbeg; kk < kend; kk += 2) {
    KK_UNROLL
    for (int P = 0; P < 2; ++P) {                      // plane kk + P computes out of value buffer P
      const int k = kk + P;
      if (k >= kend) break;
      // issued now, awaited at the end of the NEXT plane: the row words of plane k + 3, X of plane k + 3, the values of plane
      // k + 2 (addressed by the words the previous plane loaded)
      load_words(k + 3, w_stage[P], m_stage[P]);
      load_slab(k + 3, rx[P]);
      load_values(k + 2, w_stage[1 - P], m_stage[1 - P], ra[P]);
      XV yold[2] = {{0.0, 0.0}, {0.0, 0.0}}, out[2] = {{0.0, 0.0}, {0.0, 0.0}};
      if constexpr (!BETA0) load_yold(k, w_cur, yold);
      if (conforms(k, 0, w_cur[0]) || conforms(k, 1, w_cur[1])) {           // a row that does not conform computes garbage nobody stores
        const AT* av0 = abuf + ((size_t)(P * 2) * ROWS + rs) * LP;
        const AT* av1 = av0 + (size_t)ROWS * LP;
        const char* own = ring + (((2 * line) * W + ii + 1) << 7) + (XT ? 0 : c * 16); // slab line of lattice line j - 1 of the pair's first row, slot 0, this lane's piece
        // XT: the lane's piece of the X row at point ii + 1 + di sits at slot c ^ ((ii + 1 + di) & 7) (see x_dst)
        const char* own_d[3] = {own + ((c ^ (ii & 7)) << 4), own + ((c ^ ((ii + 1) & 7)) << 4), own + ((c ^ ((ii + 2) & 7)) << 4)};
        double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
        KK_UNROLL
        for (int gp = 0; gp < (NG + 1) / 2; ++gp) {                         // two groups = six positions = three 16-byte reads per row
          AV2 v0[3], v1[3];
          KK_UNROLL
          for (int i = 0; i < 3; ++i)
            if (6 * gp + 2 * i < NPOS) { v0[i] = *reinterpret_cast<const AV2*>(av0 + 6 * gp + 2 * i); v1[i] = *reinterpret_cast<const AV2*>(av1 + 6 * gp + 2 * i); }
          KK_UNROLL
          for (int gg = 0; gg < 2; ++gg) {
            const int g = 2 * gp + gg;
            if (g >= NG || (!PRES && g >= G.ng)) continue;                    // uniform
            // PRES != 0: the stencil's pattern (3 bits per group) is a compile-time constant -- straight-line code, every read of the
            // plane schedulable ahead; PRES == 0: any pattern, scalar branches
            const int e = G.e[g], pres = PRES ? (int)((PRES >> (3 * g)) & 7u) : G.pres[g];
            const int d_i = XT ? (((e >> 2) + W + 1) % W) : 0;                 // di + 1 of the group (uniform)
            const char* xb = (!XT ? own : (d_i == 0 ? own_d[0] : (d_i == 1 ? own_d[1] : own_d[2]))) + ((k + (e & 3) - 1) & 3) * SLABB + (e >> 2) * 128;
            XV x[4];
            KK_UNROLL
            for (int sl = 0; sl < 4; ++sl)                                    // X row j - 1 + sl of the group: the first row's dj = sl - 1, the second row's dj = sl - 2
              if ((sl < 3 && ((pres >> sl) & 1)) || (sl > 0 && ((pres >> (sl - 1)) & 1))) x[sl] = *reinterpret_cast<const XV*>(xb + sl * (W * 128));
            KK_UNROLL
            for (int d = 0; d < 3; ++d) {
              if (!((pres >> d) & 1)) continue;                               // the stencil has no such entry (0 * Inf would be NaN); uniform
              const int idx = 3 * gg + d;
              const double va0 = (double)v0[idx >> 1][idx & 1], va1 = (double)v1[idx >> 1][idx & 1];
              a00 = __builtin_fma(va0, x[d][0], a00);     a01 = __builtin_fma(va0, x[d][1], a01);
              a10 = __builtin_fma(va1, x[d + 1][0], a10); a11 = __builtin_fma(va1, x[d + 1][1], a11);
            }
          }
        }
        out[0][0] = alpha * a00; out[0][1] = alpha * a01; out[1][0] = alpha * a10; out[1][1] = alpha * a11;
      }
      // what the PREVIOUS plane issued: the values of plane k + 1 into the other buffer (last read in plane k - 1), X of plane
      // k + 2 into the slot that held plane k - 2 -- both behind a barrier.  This plane's stores to Y come after the wait:
      // vector-memory operations retire in order, and a wait placed after a store would wait for the store as well.
      store_values(1 - P, m_nxt, ra[1 - P]);
      store_slab(k + 2, rx[1 - P]);
      KK_UNROLL
      for (int u = 0; u < 2; ++u) {
        if (conforms(k, u, w_cur[u])) {
          double* yp = y_ptr(k, u);
          if constexpr (!BETA0) { out[u][0] += beta * yold[u][0]; out[u][1] += beta * yold[u][1]; }
          if constexpr (PART) { if (ycol1 && y_vec_ok) *reinterpret_cast<XV*>(yp) = out[u]; else { if (ycol0) yp[0] = out[u][0]; if (ycol1) yp[ys1] = out[u][1]; } }
          else if (y_vec_ok) *reinterpret_cast<XV*>(yp) = out[u]; else { yp[0] = out[u][0]; yp[ys1] = out[u][1]; }
        }
      }
      __syncthreads();
      zero_halo(k + 3);                                // its slot held plane k - 1, read for the last time before the barrier
      KK_UNROLL
      for (int u = 0; u < 2; ++u) { w_cur[u] = w_nxt[u]; w_nxt[u] = w_stage[1 - P][u]; m_nxt[u] = m_stage[1 - P][u]; }
    }
  }
}

// A row CONFORMS when its entries are, in order, a subset of the reference row's (col[q'] = r + off[q]), every entry it holds
// points inside the lattice and every entry it lacks points outside: interior rows hold them all, rows on a lattice boundary
// of a truncated stencil hold the rest, anything else (wrap-around couplings, extra or missing interior entries) does not
// conform.  arow[r] = row_map[r] and amask[r] = the entries held, or -1 and 0; *count = rows that do not conform.
template <class OffT>
__global__ __launch_bounds__(kBlock) void mv4_verify_kernel(int64_t nrows, const OffT* __restrict__ row_map, const int32_t* __restrict__ entries,
                                                            Mv4Tab offs, Mv4Tab steps, int nx, int ny, int nz, OffT* __restrict__ arow,
                                                            uint32_t* __restrict__ amask, unsigned long long* __restrict__ count) {
  // The entries of the workgroup's 256 rows are one contiguous piece of the array: it is copied into LDS with coalesced loads and every
  // work-item walks its row there (row pitch 27 words: no bank conflicts) -- a work-item reading its own row from memory, entry by
  // entry, made this kernel 6.5 ms on C3 (27e6 rows), as long as two SpMV_MV calls.  Pieces that do not fit (rows longer than the
  // stencil: they will not conform anyway) are read where they lie.
  constexpr int CAP = kBlock * kMv4MaxL;
  __shared__ int s_ent[CAP];
  // the stencil's offsets and steps are indexed by a position that differs from lane to lane: out of LDS (out of the kernel's arguments the
  // compiler serves such an index lane by lane: 2.8 ms on C3 against 1.9 with the tables in LDS)
  __shared__ long long s_off[kMv4MaxL];
  __shared__ int s_step[kMv4MaxL];
  if (threadIdx.x < kMv4MaxL) { s_off[threadIdx.x] = threadIdx.x < offs.n ? (long long)offs.e[threadIdx.x] : LLONG_MIN; s_step[threadIdx.x] = threadIdx.x < steps.n ? steps.e[threadIdx.x] : 0; }
  const int64_t r0 = (int64_t)blockIdx.x * kBlock, r = r0 + threadIdx.x;
  const int64_t rN = r0 + kBlock < nrows ? r0 + kBlock : nrows;
  const int64_t a0 = (int64_t)row_map[r0], a1 = (int64_t)row_map[rN];
  const bool staged = a1 - a0 <= CAP;                          // workgroup-uniform
  if (staged) {
    for (int64_t p = threadIdx.x; p < a1 - a0; p += kBlock) s_ent[p] = entries[a0 + p];
  }
  __syncthreads();
  if (r >= nrows) return;
  const int64_t b = (int64_t)row_map[r], len = (int64_t)row_map[r + 1] - b;
  const int i = (int)(r % nx), j = (int)((r / nx) % ny), k = (int)(r / ((int64_t)nx * ny));
  auto inside = [&](int q) {                           // steps.e[q] = (dk + 1) | (dj + 1) << 2 | (di + 1) << 4
    const int s = s_step[q], kk = k + (s & 3) - 1, jj = j + ((s >> 2) & 3) - 1, ii = i + ((s >> 4) & 3) - 1;
    return kk >= 0 && kk < nz && jj >= 0 && jj < ny && ii >= 0 && ii < nx;
  };
  bool ok = len >= 1 && len <= offs.n;
  uint32_t mask = 0;
  int q = 0;
  for (int64_t a = 0; ok && a < len; ++a) {
    const int64_t d = (int64_t)(staged ? s_ent[b - a0 + a] : entries[b + a]) - r;
    while (q < offs.n && s_off[q] != d) ++q;           // in order: the packed position of an entry is the count of held entries before it
    ok = q < offs.n && inside(q);
    if (ok) mask |= 1u << q++;
  }
  for (int z = 0; ok && z < offs.n; ++z) ok = ((mask >> z) & 1u) || !inside(z);
  arow[r]  = ok ? (OffT)b : (OffT)-1;
  amask[r] = ok ? mask : 0u;
  if (!ok) atomicAdd(count, 1ull);
}
template <class OffT>
__global__ __launch_bounds__(kBlock) void mv4_list_kernel(int64_t nrows, const OffT* __restrict__ arow, int32_t* __restrict__ list,
                                                          unsigned long long* __restrict__ cursor) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nrows && arow[r] < 0) list[atomicAdd(cursor, 1ull)] = (int32_t)r;
}

template <class OffT>
int launch_mv4_verify(int64_t nrows, const OffT* row_map, const int32_t* entries, Mv4Tab offs, Mv4Tab steps, int nx, int ny, int nz, OffT* arow, uint32_t* amask,
                      unsigned long long* count, hipStream_t st)
#ifdef KK_MV4_INSTANTIATE
{
  KK_LAUNCH((mv4_verify_kernel<OffT>), (unsigned)ceil_div(nrows, kBlock), kBlock, 0, st, nrows, row_map, entries, offs, steps, nx, ny, nz, arow, amask, count);
  return KKAMD_OK;
}
#else
;
#endif
template <class OffT>
int launch_mv4_list(int64_t nrows, const OffT* arow, int32_t* list, unsigned long long* count, hipStream_t st)
#ifdef KK_MV4_INSTANTIATE
{
  KK_LAUNCH((mv4_list_kernel<OffT>), (unsigned)ceil_div(nrows, kBlock), kBlock, 0, st, nrows, arow, list, count);
  return KKAMD_OK;
}
#else
;
#endif
#ifndef KK_MV4_INSTANTIATE
extern template int launch_mv4_verify<int32_t>(int64_t, const int32_t*, const int32_t*, Mv4Tab, Mv4Tab, int, int, int, int32_t*, uint32_t*, unsigned long long*, hipStream_t);
extern template int launch_mv4_verify<int64_t>(int64_t, const int64_t*, const int32_t*, Mv4Tab, Mv4Tab, int, int, int, int64_t*, uint32_t*, unsigned long long*, hipStream_t);
extern template int launch_mv4_list<int32_t>(int64_t, const int32_t*, int32_t*, unsigned long long*, hipStream_t);
extern template int launch_mv4_list<int64_t>(int64_t, const int64_t*, int32_t*, unsigned long long*, hipStream_t);
#endif

// The launch in three layers, so that a process loads only the code it runs (a code object is loaded whole at the first launch of one of its kernels):
//   launch_mv4_stencil<OffT, AT, NE, FL>   the twelve instantiations of the kernel for ONE stencil pattern (beta = 0 or not, X layout, partial block):
//                                          one translation unit per (types, stencil) -- kk_spmv_mv4_<types>_<stencil>.hip, 0.3 MB of code each;
//   launch_mv4_rows<OffT, AT>              the gather kernel for the rows outside the stencil, and launch_mv4_verify<OffT>, the analysis kernels:
//                                          kk_spmv_mv4_aux.hip;
//   launch_mv4<OffT, AT>                   picks the stencil (inline, instantiates nothing).
template <class OffT, class AT, int NE, unsigned FL>
int launch_mv4_stencil(const kkamd_mv4_plan* m, const kkamd_crs_t* A, const double* X, int64_t xs0, int64_t xs1, double* Y, int64_t ys0, int64_t ys1,
                       double alpha, double beta, hipStream_t st, int ncv, int ncb, bool xrow, bool xcol)
#ifdef KK_MV4_INSTANTIATE
{
  const size_t slabs = 4 * (size_t)((kMv4RJ + 2) * (kMv4RI + 2) * 128), rows = kMv4Threads / 8;
  const int yv = (ys1 == 1 && (ys0 % 2 == 0) && ((uintptr_t)Y % 16 == 0)) ? 1 : 0;
  const bool part = ncv < 16;
#ifndef KK_EMU
#define KK_MV4_ATTR(B0, XR, PT) KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_mv4_kernel<OffT, AT, NE, FL, B0, XR, PT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
#else
#define KK_MV4_ATTR(B0, XR, PT) (void)0
#endif
#define KK_MV4B(B0, XR, PT)                                                                                                        \
  do {                                                                                                                          \
    const size_t lds = slabs + 4 * rows * mv4_pitch(3 * NE + (NE & 1), (int)sizeof(AT)) * sizeof(AT);                            \
    KK_MV4_ATTR(B0, XR, PT);                                                                                                      \
    KK_LAUNCH((spmv_mv4_kernel<OffT, AT, NE, FL, B0, XR, PT>), (unsigned)(m->npi * m->npj * m->nchunk * ncb), kMv4Threads, lds, st,  \
              (const OffT*)m->d_arow, (const uint32_t*)m->d_amask, (const AT*)A->d_values, m->grp, X, xs0, xs1, Y, ys0, ys1, alpha, \
              beta, yv, m->nx, m->ny, m->nz, m->S1, m->S2, m->npi, m->npj, m->kc, ncv, ncb);                                     \
  } while (0)
  if (part) {
    if (beta == 0.0) { if (xrow && ncv % 2 == 0) KK_MV4B(true, 1, true); else if (xcol) KK_MV4B(true, 2, true); else KK_MV4B(true, 0, true); }
    else { if (xrow && ncv % 2 == 0) KK_MV4B(false, 1, true); else if (xcol) KK_MV4B(false, 2, true); else KK_MV4B(false, 0, true); }
  }
  else if (beta == 0.0) { if (xrow) KK_MV4B(true, 1, false); else if (xcol) KK_MV4B(true, 2, false); else KK_MV4B(true, 0, false); }
  else { if (xrow) KK_MV4B(false, 1, false); else if (xcol) KK_MV4B(false, 2, false); else KK_MV4B(false, 0, false); }
#undef KK_MV4B
#undef KK_MV4_ATTR
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}
#else
;
#endif

template <class OffT, class AT>
int launch_mv4_rows(const kkamd_mv4_plan* m, const kkamd_crs_t* A, const double* X, int64_t xs0, int64_t xs1, double* Y, int64_t ys0, int64_t ys1,
                    double alpha, double beta, hipStream_t st, int ncv, int ncb)
#ifdef KK_MV4_INSTANTIATE
{
  for (int q = 0; q < ncb; ++q) {
    const double* Xq = X + (int64_t)q * 16 * xs1; double* Yq = Y + (int64_t)q * 16 * ys1;
    const int ncv_q = q == ncb - 1 ? ncv : 16;
    KK_LAUNCH((mv4_rows_kernel<OffT, AT>), (unsigned)ceil_div(m->n_nc * 16, (int64_t)kBlock), kBlock, 0, st, m->n_nc, (const int32_t*)m->d_nc,
              (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, Xq, xs0, xs1, Yq, ys0, ys1, alpha, beta, ncv_q);
    KK_LAUNCH_CHECK();
  }
  return KKAMD_OK;
}
#else
;
#endif

constexpr unsigned kMv4Pat27 = 0x7FFFFFFu;              // 9 groups x {-1, 0, 1}: the 27-point stencil
constexpr unsigned kMv4Pat7  = 2u | 2u << 3 | 7u << 6 | 2u << 9 | 2u << 12;   // (dk, di) = (-1,0) (0,-1) (0,0) (0,1) (1,0): the 7-point stencil
#define KK_MV4_FOR_ALL(F) F(int32_t, double) F(int32_t, float) F(int64_t, double) F(int64_t, float)
#define KK_MV4_DECL(OT, AT_)                                                                                                                                \
  extern template int launch_mv4_stencil<OT, AT_, 9, kMv4Pat27>(const kkamd_mv4_plan*, const kkamd_crs_t*, const double*, int64_t, int64_t, double*, int64_t, int64_t, double, double, hipStream_t, int, int, bool, bool); \
  extern template int launch_mv4_stencil<OT, AT_, 5, kMv4Pat7>(const kkamd_mv4_plan*, const kkamd_crs_t*, const double*, int64_t, int64_t, double*, int64_t, int64_t, double, double, hipStream_t, int, int, bool, bool);  \
  extern template int launch_mv4_stencil<OT, AT_, 5, 0u>(const kkamd_mv4_plan*, const kkamd_crs_t*, const double*, int64_t, int64_t, double*, int64_t, int64_t, double, double, hipStream_t, int, int, bool, bool);        \
  extern template int launch_mv4_stencil<OT, AT_, 9, 0u>(const kkamd_mv4_plan*, const kkamd_crs_t*, const double*, int64_t, int64_t, double*, int64_t, int64_t, double, double, hipStream_t, int, int, bool, bool);        \
  extern template int launch_mv4_rows<OT, AT_>(const kkamd_mv4_plan*, const kkamd_crs_t*, const double*, int64_t, int64_t, double*, int64_t, int64_t, double, double, hipStream_t, int, int);
#ifndef KK_MV4_INSTANTIATE
KK_MV4_FOR_ALL(KK_MV4_DECL)
#endif
#undef KK_MV4_DECL

template <class OffT, class AT>
inline int launch_mv4(const kkamd_spmv_plan* plan, const kkamd_crs_t* A, const double* X, int64_t xs0, int64_t xs1, double* Y, int64_t ys0, int64_t ys1,
                      double alpha, double beta, hipStream_t st, int ncv = 16, int ncb = 1) {
  const kkamd_mv4_plan* m = plan->mv4;
  const bool xrow = xs1 == 1 && (xs0 % 2 == 0) && ((uintptr_t)X % 16 == 0);
  const bool xcol = xs0 == 1 && plan->tune.mv4_xcol;   // column-major X: the column-wise piece order with swizzled slab rows
  unsigned pat = 0;                                    // 3 bits per group: which of dj = -1, 0, 1 it holds
  for (int g = 0; g < m->grp.ng; ++g) pat |= (unsigned)m->grp.pres[g] << (3 * g);
  int rc;
  if (m->grp.ng == 9 && pat == kMv4Pat27) rc = launch_mv4_stencil<OffT, AT, 9, kMv4Pat27>(m, A, X, xs0, xs1, Y, ys0, ys1, alpha, beta, st, ncv, ncb, xrow, xcol);
  else if (m->grp.ng == 5 && pat == kMv4Pat7) rc = launch_mv4_stencil<OffT, AT, 5, kMv4Pat7>(m, A, X, xs0, xs1, Y, ys0, ys1, alpha, beta, st, ncv, ncb, xrow, xcol);
  else if (m->grp.ng <= 5) rc = launch_mv4_stencil<OffT, AT, 5, 0u>(m, A, X, xs0, xs1, Y, ys0, ys1, alpha, beta, st, ncv, ncb, xrow, xcol);
  else rc = launch_mv4_stencil<OffT, AT, 9, 0u>(m, A, X, xs0, xs1, Y, ys0, ys1, alpha, beta, st, ncv, ncb, xrow, xcol);
  if (rc) return rc;
  if (m->n_nc > 0) return launch_mv4_rows<OffT, AT>(m, A, X, xs0, xs1, Y, ys0, ys1, alpha, beta, st, ncv, ncb);
  return KKAMD_OK;
}

}  // namespace kk
