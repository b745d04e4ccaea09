// kk_spmv_struct.hip -- KokkosSparse::Experimental::spmv_struct for gfx950: SpMV on a CRS matrix that is known to
// come from a 3/5/9/7/27-point stencil on an ni x nj x nk grid.
//
// Reference (sparse/impl/KokkosSparse_spmv_struct_impl.hpp): interior grid points never read entries(): the column of
// the idx-th value of an interior row is row + columnOffsets(idx) (:236-242, :270-278, :308-320, :350-360, :392-420),
// y(row) = beta*y(row) + alpha*sum (:264; here y is not read at all when beta == 0, the BLAS convention the reference's
// CRS kernels follow -- the result differs from :264 only where the incoming y holds Inf/NaN); exterior points take the ordinary CRS row (:508-618); modes T/H ignore the
// structure altogether (:733-773).  On a GPU the reference maps ONE interior row to a few vector lanes and gathers x
// through the texture path, like its CRS kernel.
//
// gfx950-native structure: 8 bytes per nonzero instead of 12, and NO gather at all --
//   * a workgroup takes a run of R = 64 or 128 consecutive interior rows of one grid line ("pencil" along i).  Their values
//     are one contiguous block of R*S doubles in values(): it is streamed into LDS with coalesced loads;
//   * the x entries those rows need are NL grid lines (1 / 3 / 3 / 5 / 9 for 3/5/9/7/27-pt) of R+2 consecutive
//     elements each: staged in LDS with coalesced loads as well (9.4 KB for 27-pt), every element reused up to 27 times;
//   * two work-items per row walk the stencil out of LDS (values at stride S, x at unit stride across lanes), one
//     shuffle combines them, y is written coalesced.
//   The contiguity of the values block is CHECKED per workgroup from row_map (the reference reads row_map per row too);
//   a run whose rows do not have exactly S entries each falls back to reading values through row_map.
//   * exterior rows (2.4 % of the rows at 300^3) are enumerated arithmetically and take 8 lanes per CRS row.
#include "kk_common.h"
#include <climits>

namespace kk {

constexpr int kMaxStencil = 27, kMaxLines = 9;

struct StencilDesc {
  int ndim, S, NL;
  int64_t ni, nj, nk;
  int line[kMaxStencil];       // idx -> which staged x line
  int di[kMaxStencil];         // idx -> offset along i
  int64_t line_off[kMaxLines]; // line -> offset of its row relative to the current row (multiple of ni)
};

int g_struct_remap = 0;
int g_struct_group = 0;
int g_struct_strip = 0;
int g_struct_lds_pad_kb = 0;

static int make_stencil(int stencil_type, int ndim, const int64_t* st, StencilDesc* d) {
  d->ndim = ndim; d->ni = st[0]; d->nj = ndim > 1 ? st[1] : 1; d->nk = ndim > 2 ? st[2] : 1;
  const int64_t ni = d->ni, nj = d->nj;
  int n = 0;
  if (ndim == 1) {
    d->NL = 1; d->line_off[0] = 0;
    for (int i = -1; i <= 1; ++i) { d->line[n] = 0; d->di[n] = i; ++n; }
  } else if (ndim == 2 && stencil_type == 1) {          // -ni, -1, 0, 1, ni
    d->NL = 3; d->line_off[0] = -ni; d->line_off[1] = 0; d->line_off[2] = ni;
    d->line[0] = 0; d->di[0] = 0;
    for (int i = -1; i <= 1; ++i) { d->line[1 + (i + 1)] = 1; d->di[1 + (i + 1)] = i; }
    d->line[4] = 2; d->di[4] = 0; n = 5;
  } else if (ndim == 2) {
    d->NL = 3;
    for (int j = -1; j <= 1; ++j) {
      d->line_off[j + 1] = j * ni;
      for (int i = -1; i <= 1; ++i) { d->line[n] = j + 1; d->di[n] = i; ++n; }
    }
  } else if (stencil_type == 1) {                        // -ni*nj, -ni, -1, 0, 1, ni, ni*nj
    d->NL = 5;
    d->line_off[0] = -ni * nj; d->line_off[1] = -ni; d->line_off[2] = 0; d->line_off[3] = ni; d->line_off[4] = ni * nj;
    d->line[0] = 0; d->di[0] = 0; d->line[1] = 1; d->di[1] = 0;
    for (int i = -1; i <= 1; ++i) { d->line[2 + (i + 1)] = 2; d->di[2 + (i + 1)] = i; }
    d->line[5] = 3; d->di[5] = 0; d->line[6] = 4; d->di[6] = 0; n = 7;
  } else {
    d->NL = 9;
    for (int k = -1; k <= 1; ++k)
      for (int j = -1; j <= 1; ++j) {
        const int l   = (k + 1) * 3 + (j + 1);
        d->line_off[l] = k * ni * nj + j * ni;
        for (int i = -1; i <= 1; ++i) { d->line[n] = l; d->di[n] = i; ++n; }
      }
  }
  d->S = n;
  return KKAMD_OK;
}

// compile-time stencil tables: idx -> (staged x line, offset along i); line -> (dj, dk)
template <int NDIM, int ST> struct Stencil;
template <int ST> struct Stencil<1, ST> {
  static constexpr int S = 3, NL = 1;
  static constexpr int line(int) { return 0; }
  static constexpr int di(int idx) { return idx - 1; }
  static constexpr int dj(int) { return 0; }
  static constexpr int dk(int) { return 0; }
};
template <> struct Stencil<2, 1> {
  static constexpr int S = 5, NL = 3;
  static constexpr int line(int idx) { return idx == 0 ? 0 : (idx == 4 ? 2 : 1); }
  static constexpr int di(int idx) { return (idx >= 1 && idx <= 3) ? idx - 2 : 0; }
  static constexpr int dj(int l) { return l - 1; }
  static constexpr int dk(int) { return 0; }
};
template <> struct Stencil<2, 2> {
  static constexpr int S = 9, NL = 3;
  static constexpr int line(int idx) { return idx / 3; }
  static constexpr int di(int idx) { return idx % 3 - 1; }
  static constexpr int dj(int l) { return l - 1; }
  static constexpr int dk(int) { return 0; }
};
template <> struct Stencil<3, 1> {
  static constexpr int S = 7, NL = 5;
  static constexpr int line(int idx) { return idx <= 1 ? idx : (idx <= 4 ? 2 : idx - 2); }
  static constexpr int di(int idx) { return (idx >= 2 && idx <= 4) ? idx - 3 : 0; }
  static constexpr int dj(int l) { return l == 1 ? -1 : (l == 3 ? 1 : 0); }
  static constexpr int dk(int l) { return l == 0 ? -1 : (l == 4 ? 1 : 0); }
};
template <> struct Stencil<3, 2> {
  static constexpr int S = 27, NL = 9;
  static constexpr int line(int idx) { return idx / 3; }
  static constexpr int di(int idx) { return idx % 3 - 1; }
  static constexpr int dj(int l) { return l % 3 - 1; }
  static constexpr int dk(int l) { return l / 3 - 1; }
};

// interior rows: blockIdx.x -> (pencil, chunk of R rows along i), R = 64 or 128 picked per launch for the better fill of
// the last chunk of a grid line.  Work-item t of 2R: row r = t mod R, stencil half h = t / R (wave-uniform, so both halves
// are straight-line code over compile-time tables).
template <class OffT, class AT, class YT, int NDIM, int ST, int R>
__global__ __launch_bounds__(2 * R) void spmv_struct_interior_kernel(int64_t ni, int64_t nj, int chunks_per_pencil,
                                                                      const OffT* __restrict__ rm, const AT* __restrict__ val,
                                                                      const YT* __restrict__ x, YT* __restrict__ y, YT alpha,
                                                                      YT beta, int remap, int nkm) {
  using St = Stencil<NDIM, ST>;
  constexpr int S = St::S, NL = St::NL, NT = 2 * R;
  constexpr int HS  = (S + 1) / 2;                             // first stencil half
  __shared__ AT s_v[R * S];
  __shared__ YT s_x[NL][R + 2];
  __shared__ YT s_part[R];
  const int t = threadIdx.x;
  // 32-bit index arithmetic on purpose: 64-bit divisions cost ~100 scalar instructions each at the start of every wave
  unsigned bid = blockIdx.x;
  if (remap >> 8) bid = (unsigned)xcd_group_order(bid, gridDim.x, remap >> 8);
  if (remap & 1) {  // XCD-contiguous order (knob struct_remap): workgroup b runs on XCD b % 8, give each XCD one slab of the grid
    const unsigned q = gridDim.x / kNumXcd, rem = gridDim.x % kNumXcd, xc = bid % kNumXcd;
    bid = xc * q + (xc < rem ? xc : rem) + bid / kNumXcd;
  }
  unsigned pencil = bid / (unsigned)chunks_per_pencil;
  int chunk       = (int)(bid - pencil * (unsigned)chunks_per_pencil);
  int64_t j = 0, k = 0;
  const unsigned JL = (unsigned)(remap >> 16) & 0xffu;
  if (NDIM >= 2 && JL) {
    // Strip order (knob struct_strip): workgroup b runs on XCD b % 8; that XCD owns, inside every block of 8*JL grid lines,
    // the JL consecutive lines of its strip and walks them plane after plane (chunk fastest, then line, then k).  The x
    // lines of planes k and k+1 it just used are still in ITS L2 when it gets to plane k+1, so x crosses the fabric about
    // once instead of once per XCD, while the eight XCDs together still sweep one 8*JL-line region of one plane at a time.
    const unsigned xc = bid & (kNumXcd - 1), seq = bid >> 3, cpp = (unsigned)chunks_per_pencil;
    const unsigned q1 = seq / cpp, q2 = q1 / JL, jb = q2 / (unsigned)nkm;
    chunk = (int)(seq - q1 * cpp);
    const unsigned jq = (jb * kNumXcd + xc) * JL + (q1 - q2 * JL), kq = q2 - jb * (unsigned)nkm;
    if (jq >= (unsigned)(nj - 2)) return;                      // the last block of lines is padded: the whole workgroup leaves
    j = (int64_t)jq + 1;
    k = (NDIM == 3) ? (int64_t)kq + 1 : 0;
  } else if (NDIM == 2) j = (int64_t)pencil + 1;
  else if (NDIM == 3) { const unsigned njm = (unsigned)(nj - 2), kq = pencil / njm; k = (int64_t)kq + 1; j = (int64_t)(pencil - kq * njm) + 1; }
  const int64_t i0   = 1 + (int64_t)chunk * R;                 // first interior i of this chunk
  const int nr       = (int)((ni - 1 - i0 < R) ? ni - 1 - i0 : R);
  const int64_t row0 = (k * nj + j) * ni + i0;
  const int r = t & (R - 1), h = t / R;
  // everything that does not depend on row_map is requested first: the x lines and the old y
  // the NL lines of R+2 x entries are dealt flat over the workgroup (element e -> line e / (R+2)), so every load
  // instruction runs with full waves: ceil(NL*(R+2) / 2R) loads per work-item instead of NL with R+2 active lanes
  constexpr int XW = R + 2, XN = NL * XW, XU = (XN + NT - 1) / NT;
  YT xl[XU];
  KK_UNROLL
  for (int u = 0; u < XU; ++u) {
    const int e = u * NT + t, l = e / XW, pos = e - l * XW;
    const int64_t off = ((int64_t)St::dk(l) * nj + St::dj(l)) * ni;
    xl[u] = (e < XN && pos < nr + 2 && !(remap & 8)) ? x[row0 - 1 + off + pos] : YT(0);
  }
  YT yold = (h == 0 && r < nr && beta != YT(0)) ? y[row0 + r] : YT(0);     // beta == 0: y is write-only (NaN-safe)
  const long long v0    = (long long)rm[row0];
  const bool contiguous = (long long)rm[row0 + nr] - v0 == (long long)nr * S;     // rows of exactly S entries each
  if (contiguous) {
    // the block of nr*S values, two per load (a pair starts at an even element of the block, i.e. at an 8-byte --
    // not necessarily 16-byte -- aligned address: vec2 carries that alignment so one 128-bit load is emitted anyway)
    constexpr int VP2 = ((R * S + 1) / 2 + NT - 1) / NT;
    struct vec2 { AT x, y; };                                    // alignof(vec2) == sizeof(AT)
    vec2 vv[VP2];
    const int nval = nr * S, npair = nval >> 1;
    KK_UNROLL
    for (int u = 0; u < VP2; ++u) {
      const int q = u * NT + t;
      if (q < npair) vv[u] = *reinterpret_cast<const vec2*>(val + v0 + 2 * q);
      else { vv[u].x = (2 * q < nval) ? val[v0 + 2 * q] : AT(0); vv[u].y = AT(0); }      // the odd last value, if any
    }
    KK_UNROLL
    for (int u = 0; u < VP2; ++u) { const int q = u * NT + t; if (q < (R * S + 1) / 2) { s_v[2 * q] = vv[u].x; if (2 * q + 1 < R * S) s_v[2 * q + 1] = vv[u].y; } }
  }
  KK_UNROLL
  for (int u = 0; u < XU; ++u) { const int e = u * NT + t; if (e < XN) (&s_x[0][0])[e] = xl[u]; }
  __syncthreads();
  YT sum = YT(0);
  if (remap & 16) sum = (YT)s_v[t];
  else if (r < nr) {
    if (contiguous) {
      if (h == 0) {
        KK_UNROLL
        for (int idx = 0; idx < HS; ++idx) sum += (YT)s_v[r * S + idx] * s_x[St::line(idx)][r + 1 + St::di(idx)];
      } else {
        KK_UNROLL
        for (int idx = HS; idx < S; ++idx) sum += (YT)s_v[r * S + idx] * s_x[St::line(idx)][r + 1 + St::di(idx)];
      }
    } else {                                                   // general row_map: values straight from HBM
      const long long ro = (long long)rm[row0 + r];
      if (h == 0) {
        KK_UNROLL
        for (int idx = 0; idx < HS; ++idx) sum += (YT)val[ro + idx] * s_x[St::line(idx)][r + 1 + St::di(idx)];
      } else {
        KK_UNROLL
        for (int idx = HS; idx < S; ++idx) sum += (YT)val[ro + idx] * s_x[St::line(idx)][r + 1 + St::di(idx)];
      }
    }
  }
  if (h == 1) s_part[r] = sum;
  __syncthreads();
  if (h == 0 && r < nr && (!(remap & 2) || sum == YT(1.2345e30))) y[row0 + r] = (beta == YT(0)) ? alpha * (sum + s_part[r]) : beta * yold + alpha * (sum + s_part[r]);
}

// exterior rows, 8 lanes per CRS row.  e -> row: bottom plane, then for every middle plane the j = 0 line, the
// j = nj-1 line and the two end points of the lines between, then the top plane (1-D / 2-D: the same with fewer levels).
__device__ __forceinline__ int64_t exterior_row(const StencilDesc& d, int64_t e) {
  const int64_t ni = d.ni, nj = d.nj, nk = d.nk;
  if (d.ndim == 1) return e * (ni - 1);
  const int64_t per_plane = 2 * ni + 2 * (nj - 2);          // exterior points of one 2-D layer
  int64_t kk2 = 0, r = e;
  if (d.ndim == 3) {
    const int64_t plane = ni * nj;
    if (e < plane) return e;
    const int64_t mid = (nk - 2) * per_plane;
    if (e >= plane + mid) return (nk - 1) * plane + (e - plane - mid);
    kk2 = 1 + (e - plane) / per_plane;
    r   = (e - plane) % per_plane;
  }
  const int64_t base = kk2 * ni * nj;
  if (r < ni) return base + r;
  if (r < 2 * ni) return base + (nj - 1) * ni + (r - ni);
  const int64_t r2 = r - 2 * ni;
  return base + (1 + r2 / 2) * ni + ((r2 & 1) ? ni - 1 : 0);
}
template <class OffT, class AT, class YT>
__global__ __launch_bounds__(kBlock) void spmv_struct_exterior_kernel(StencilDesc d, int64_t num_ext, const OffT* __restrict__ rm,
                                                                      const int32_t* __restrict__ ent, const AT* __restrict__ val,
                                                                      const YT* __restrict__ x, YT* __restrict__ y, YT alpha,
                                                                      YT beta) {
  const int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 8;
  const int lane  = threadIdx.x & 7;
  int64_t row = 0;
  YT sum      = YT(0);
  if (e < num_ext) {
    row = exterior_row(d, e);
    for (int64_t q = (int64_t)rm[row] + lane; q < (int64_t)rm[row + 1]; q += 8) sum += (YT)val[q] * x[ent[q]];
  }
  sum = group_sum(sum, 8);
  if (e < num_ext && lane == 0) y[row] = (beta == YT(0)) ? alpha * sum : beta * y[row] + alpha * sum;
}

template <class OffT, class AT, class YT>
static int spmv_struct_typed(const StencilDesc& d, const kkamd_crs_t* A, double alpha, const void* x_, double beta, void* y_,
                             hipStream_t st) {
  const OffT* rm = (const OffT*)A->d_row_map;
  const AT* val  = (const AT*)A->d_values;
  const YT* x    = (const YT*)x_;
  YT* y          = (YT*)y_;
  const int64_t ni = d.ni, nj = d.nj, nk = d.nk;
  const int64_t pencils  = d.ndim == 1 ? 1 : d.ndim == 2 ? nj - 2 : (nj - 2) * (nk - 2);
  const int64_t interior = ni - 2;
  int64_t num_int = 0;
  if (interior > 0 && pencils > 0) {
    // rows per workgroup: whichever of 64 / 128 leaves less of a grid line's last chunk empty (ties: 128)
    const int64_t c128 = ceil_div(interior, 128), c64 = ceil_div(interior, 64);
    const bool use64  = c64 * 64 < c128 * 128;
    const int64_t cpp = use64 ? c64 : c128;
    if (pencils * cpp > INT32_MAX) return fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_struct: grid too large");
    num_int = interior * pencils;
    // strip order (see the kernel): the lines are padded to whole blocks of 8 * strip
    const int strip    = (d.ndim >= 2 && nj - 2 >= 8 * g_struct_strip) ? (g_struct_strip & 0xff) : 0;
    const int64_t nkm  = d.ndim == 3 ? nk - 2 : 1;
    const int64_t nwg  = strip ? ceil_div(nj - 2, (int64_t)kNumXcd * strip) * kNumXcd * strip * nkm * cpp : pencils * cpp;
    if (nwg > INT32_MAX) return fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_struct: grid too large");
    const int flags    = g_struct_remap | (g_struct_group << 8) | (strip << 16);
#define KK_STRUCT_LAUNCH(ND, STT)                                                                                          \
  do {                                                                                                                     \
    if (use64) KK_LAUNCH((spmv_struct_interior_kernel<OffT, AT, YT, ND, STT, 64>), (unsigned)nwg, 128, (size_t)g_struct_lds_pad_kb * 1024, st, ni, nj, \
                         (int)cpp, rm, val, x, y, (YT)alpha, (YT)beta, flags, (int)nkm);                                   \
    else KK_LAUNCH((spmv_struct_interior_kernel<OffT, AT, YT, ND, STT, 128>), (unsigned)nwg, 256, (size_t)g_struct_lds_pad_kb * 1024, st, ni, nj,      \
                   (int)cpp, rm, val, x, y, (YT)alpha, (YT)beta, flags, (int)nkm);                                         \
  } while (0)
    if (d.ndim == 1) KK_STRUCT_LAUNCH(1, 1);
    else if (d.ndim == 2 && d.S == 5) KK_STRUCT_LAUNCH(2, 1);
    else if (d.ndim == 2) KK_STRUCT_LAUNCH(2, 2);
    else if (d.S == 7) KK_STRUCT_LAUNCH(3, 1);
    else KK_STRUCT_LAUNCH(3, 2);
#undef KK_STRUCT_LAUNCH
  }
  const int64_t num_ext = ni * nj * nk - num_int;
  if (num_ext > 0)
    KK_LAUNCH((spmv_struct_exterior_kernel<OffT, AT, YT>), (unsigned)ceil_div(num_ext * 8, kBlock), kBlock, 0, st, d, num_ext, rm,
              (const int32_t*)A->d_entries, val, x, y, (YT)alpha, (YT)beta);
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

}  // namespace kk

extern "C" int kkamd_spmv_struct(const kkamd_crs_t* A, char mode, int stencil_type, int ndim, const int64_t* structure,
                                 double alpha, const void* d_x, double beta, void* d_y, int vector_type,
                                 kkamd_stream_t stream) {
  int rc = kk::check_crs(A);
  if (rc) return rc;
  bool trans = false;
  if ((rc = kk::parse_mode(mode, &trans))) return rc;
  // transpose modes ignore the structure (spmv_struct_beta_transpose, :733-773)
  if (trans) return kkamd_spmv(nullptr, A, mode, alpha, d_x, beta, d_y, vector_type, stream);
  if (A->num_rows <= 0) return KKAMD_OK;                       // :659-661
  if (ndim < 1 || ndim > 3 || !structure) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_struct: structure must have 1..3 extents");
  if (stencil_type != 1 && stencil_type != 2) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_struct: stencil_type must be 1 (FD) or 2 (FE)");
  int64_t n = 1;
  for (int q = 0; q < ndim; ++q) {
    if (structure[q] < 2) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_struct: every grid extent must be at least 2");
    n *= structure[q];
  }
  if (n != A->num_rows || A->num_cols < A->num_rows)
    return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_struct: structure (%lld points) does not match the %lld x %lld matrix",
                    (long long)n, (long long)A->num_rows, (long long)A->num_cols);
  if (vector_type != KKAMD_F32 && vector_type != KKAMD_F64)
    return kk::fail(KKAMD_ERR_UNSUPPORTED, "kkamd_spmv_struct: unsupported vector_type %d", vector_type);
  if (!d_x || !d_y) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_struct: null vector");
  kk::StencilDesc d;
  kk::make_stencil(stencil_type, ndim, structure, &d);
  hipStream_t st = kk::to_hip(stream);
  const bool o64 = A->offset_type == KKAMD_I64;
  if (A->value_type == KKAMD_F64 && vector_type == KKAMD_F64)
    return o64 ? kk::spmv_struct_typed<int64_t, double, double>(d, A, alpha, d_x, beta, d_y, st)
               : kk::spmv_struct_typed<int32_t, double, double>(d, A, alpha, d_x, beta, d_y, st);
  if (A->value_type == KKAMD_F32 && vector_type == KKAMD_F32)
    return o64 ? kk::spmv_struct_typed<int64_t, float, float>(d, A, alpha, d_x, beta, d_y, st)
               : kk::spmv_struct_typed<int32_t, float, float>(d, A, alpha, d_x, beta, d_y, st);
  if (A->value_type == KKAMD_F32 && vector_type == KKAMD_F64)
    return o64 ? kk::spmv_struct_typed<int64_t, float, double>(d, A, alpha, d_x, beta, d_y, st)
               : kk::spmv_struct_typed<int32_t, float, double>(d, A, alpha, d_x, beta, d_y, st);
  return kk::fail(KKAMD_ERR_UNSUPPORTED, "kkamd_spmv_struct: unsupported (value,vector) type pair (%d,%d)", A->value_type, vector_type);
}
