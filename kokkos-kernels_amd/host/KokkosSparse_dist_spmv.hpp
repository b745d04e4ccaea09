// KokkosSparse::Experimental::DistributedSpMV -- C++ face of the multi-GPU SpMV (kkamd_dist_spmv_*, include/kkamd.h).
// The reference has no distributed layer (upstream that is Tpetra's job): this is the host-side wrapper the
// north star asks for ("host code stays C++"), shaped like the other handles: construct once per (matrix slab, partition),
// apply per SpMV.  One process per GPU; rank r owns rows [row_offsets[r], row_offsets[r+1]) of A (local row_map, GLOBAL
// column indices), the matching slab of y and shard of x.
#pragma once
#include <vector>
#include "KokkosSparse_CrsMatrix.hpp"
#include "KokkosSparse_spmv.hpp"
#include "kkamd_status.hpp"

namespace KokkosSparse { namespace Experimental {

enum DistExchange { DIST_EXCHANGE_AUTO = 0, DIST_EXCHANGE_HALO = 1, DIST_EXCHANGE_ALLGATHER = 2, DIST_EXCHANGE_ALLGATHER_P2P = 3, DIST_EXCHANGE_HALO_SET = 4,
                    DIST_EXCHANGE_ALLGATHER_TIMED = 5 /* the form of the all-gather is chosen by timing at creation (include/kkamd.h) */ };

template <class AMatrix>
class DistributedSpMV {
 public:
  using value_type = typename AMatrix::non_const_value_type;
  // rccl_id: the 128 bytes rank 0 obtained from unique_id(), handed to every rank by the host's launcher (MPI_Bcast, ...);
  // transport: a caller-supplied kkamd_transport_t instead of the built-in RCCL (then rccl_id may be null)
  DistributedSpMV(const Kokkos::HIP& space, const AMatrix& A_local, const std::vector<int64_t>& row_offsets, int rank, const void* rccl_id,
                  SPMVAlgorithm algo = SPMV_DEFAULT, DistExchange exchange = DIST_EXCHANGE_AUTO, bool overlap = true,
                  const kkamd_transport_t* transport = nullptr)
      : A_(A_local) {
    kkamd_crs_t d = KokkosSparse::Impl::make_crs_desc(A_local);
    KokkosSparse::Impl::kkamd_check(kkamd_dist_spmv_create(&op_, &d, row_offsets.data(), (int)row_offsets.size() - 1, rank, rccl_id, transport, (int)algo,
                                             (int)exchange, overlap ? 1 : 0, std::is_same<value_type, double>::value ? KKAMD_F64 : KKAMD_F32,
                                             reinterpret_cast<kkamd_stream_t>(space.hip_stream())));
  }
  ~DistributedSpMV() { if (op_) kkamd_dist_spmv_destroy(op_); }
  DistributedSpMV(const DistributedSpMV&)            = delete;
  DistributedSpMV& operator=(const DistributedSpMV&) = delete;

  static void unique_id(void* id128) { KokkosSparse::Impl::kkamd_check(kkamd_dist_unique_id(id128)); }

  // the rank's own window of the operator's full-length x: an x kept there is never copied by apply()
  value_type* x_local() const { void* p = nullptr; KokkosSparse::Impl::kkamd_check(kkamd_dist_spmv_x_local(op_, &p, nullptr)); return static_cast<value_type*>(p); }

  // y_shard := alpha * A_local * exchanged(x) + beta * y_shard
  template <class XVector, class YVector>
  void apply(const Kokkos::HIP& space, value_type alpha, const XVector& x_shard, value_type beta, const YVector& y_shard) const {
    KokkosSparse::Impl::kkamd_check(kkamd_dist_spmv_apply(op_, (double)alpha, x_shard.data(), (double)beta, y_shard.data(), 0,
                                            reinterpret_cast<kkamd_stream_t>(space.hip_stream())));
  }
  int64_t query(const char* key) const { int64_t v = 0; KokkosSparse::Impl::kkamd_check(kkamd_dist_spmv_query(op_, key, &v)); return v; }

 private:
  AMatrix A_;                         // keeps the slab's views alive
  kkamd_dist_spmv_t* op_ = nullptr;
};

}}  // namespace KokkosSparse::Experimental
