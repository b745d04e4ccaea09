// KokkosSparse_IOUtils.hpp -- the matrix file formats of the reference behind its own function names:
//   KokkosSparse::Impl::read_mtx, read_kokkos_crst_matrix, write_kokkos_crst_matrix
//   (sparse/src/KokkosSparse_IOUtils.hpp:785-987, 1238-1290, 741-783; .bin / .crs layouts :488-520, 681-738).
// Host code over the Kokkos stand-in (Kokkos_Shim.hpp); the arrays are assembled on the host and copied into the
// matrix's memory space.  Real / integer / pattern MatrixMarket fields (the hot path is real-valued); Harwell-Boeing
// is not provided.  Same reader semantics as kokkos-kernels_amd/io.py, which documents them in one place.
#pragma once
#include <algorithm>
#include <cstdint>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>
#include "KokkosSparse_CrsMatrix.hpp"

namespace KokkosSparse {
namespace Impl {

inline bool kkamd_endswith(const std::string& s, const std::string& suffix) {
  return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

// CRS arrays of a MatrixMarket file.  symmetrize / remove_diagonal / transpose as in the reference (:786).
template <typename lno_t, typename size_type, typename scalar_t>
void read_mtx(const char* fileName, lno_t* nrows, lno_t* ncols, std::vector<size_type>& xadj, std::vector<lno_t>& adj,
              std::vector<scalar_t>& ew, bool symmetrize = false, bool remove_diagonal = true, bool transpose = false) {
  std::ifstream in(fileName);
  if (!in.is_open()) throw std::runtime_error("File cannot be opened\n");
  std::string banner;
  std::getline(in, banner);
  if (banner.size() < 2 || banner[0] != '%' || banner[1] != '%') throw std::runtime_error("Invalid MM file. Line-1\n");
  auto has = [&](const char* w) { return banner.find(w) != std::string::npos; };
  if (!has("matrix")) {
    if (has("vector")) throw std::runtime_error("MatrixMarket \"vector\" is not supported by KokkosKernels read_mtx()");
    throw std::runtime_error("MatrixMarket file header is missing the object type.");
  }
  const bool coordinate = has("coordinate"), array = !coordinate && has("array");
  enum { NOFIELD, REAL, INTEGER, PATTERN } field = NOFIELD;
  if (has("real") || has("double")) field = REAL;
  else if (has("complex")) throw std::runtime_error("scalar_t in read_mtx() incompatible with complex-typed MatrixMarket file.");
  else if (has("integer")) field = INTEGER;
  else if (has("pattern")) field = PATTERN;
  enum { NOSYM, GENERAL, SYMMETRIC, SKEW, HERMITIAN } sym = NOSYM;
  if (has("general")) sym = GENERAL;
  else if (has("skew-symmetric")) sym = SKEW;
  else if (has("symmetric")) sym = SYMMETRIC;
  else if (has("hermitian") || has("Hermitian")) sym = HERMITIAN;
  if (array) {
    if (sym == NOSYM) sym = GENERAL;
    if (sym != GENERAL)
      throw std::runtime_error("array format MatrixMarket file must have general symmetry (optional to include \"general\")");
  }
  if (!coordinate && !array) throw std::runtime_error("MatrixMarket file header is missing the format.");
  if (field == NOFIELD) throw std::runtime_error("MatrixMarket file header is missing the field type.");
  if (sym == NOSYM) throw std::runtime_error("MatrixMarket file header is missing the symmetry type.");
  std::string line;
  do { std::getline(in, line); } while (!line.empty() && line[0] == '%');
  long long nr = 0, nc = 0, nnz = 0;
  {
    std::istringstream ss(line);
    ss >> nr >> nc;
    if (coordinate) ss >> nnz; else nnz = nr * nc;
  }
  symmetrize = symmetrize || sym != GENERAL;
  if (symmetrize && nr != nc) throw std::runtime_error("A non-square matrix cannot be symmetrized.");
  if (array && symmetrize) throw std::runtime_error("array format MatrixMarket file cannot be symmetrized.");
  if (array && field == PATTERN) throw std::runtime_error("array format MatrixMarket file can't have \"pattern\" field type.");
  // (row, column, position in the reference's edge list, value): the position keeps ties in append order
  std::vector<std::tuple<long long, long long, long long, scalar_t>> edges;
  edges.reserve((size_t)(symmetrize ? 2 * nnz : nnz));
  for (long long i = 0; i < nnz; ++i) {
    if (!std::getline(in, line)) throw std::runtime_error("MatrixMarket file ends before all entries were read");
    std::istringstream ss(line);
    long long s, d;
    double w = 1;
    if (array) { s = i % nr + 1; d = i / nr + 1; } else ss >> s >> d;
    if (field != PATTERN) ss >> w;
    long long src = s - 1, dst = d - 1;
    if (transpose) std::swap(src, dst);
    if (src == dst) {
      if (!remove_diagonal) edges.emplace_back(src, dst, 2 * i, (scalar_t)w);
      continue;
    }
    edges.emplace_back(src, dst, 2 * i, (scalar_t)w);
    if (symmetrize) edges.emplace_back(dst, src, 2 * i + 1, (scalar_t)(sym == SKEW ? -w : w));
  }
  std::sort(edges.begin(), edges.end(), [](const auto& a, const auto& b) {
    return std::tie(std::get<0>(a), std::get<1>(a), std::get<2>(a)) < std::tie(std::get<0>(b), std::get<1>(b), std::get<2>(b));
  });
  if (transpose) std::swap(nr, nc);
  *nrows = (lno_t)nr; *ncols = (lno_t)nc;
  xadj.assign((size_t)nr + 1, 0); adj.clear(); ew.clear();
  size_t e = 0;
  for (long long r = 0; r < nr; ++r) {
    xadj[(size_t)r] = (size_type)adj.size();
    bool first = true;
    for (; e < edges.size() && std::get<0>(edges[e]) == r; ++e) {
      // when symmetrizing, a repeated (row, column) pair collapses to its first occurrence (:976-981)
      if (first || !symmetrize || std::get<1>(edges[e - 1]) != std::get<1>(edges[e])) {
        adj.push_back((lno_t)std::get<1>(edges[e])); ew.push_back(std::get<3>(edges[e]));
      }
      first = false;
    }
  }
  xadj[(size_t)nr] = (size_type)adj.size();
}

template <typename crsMat_t>
crsMat_t read_kokkos_crst_matrix(const char* filename_) {
  using graph_t   = typename crsMat_t::StaticCrsGraphType;
  using rowmap_t  = typename graph_t::row_map_type::non_const_type;
  using cols_t    = typename graph_t::entries_type::non_const_type;
  using values_t  = typename crsMat_t::values_type::non_const_type;
  using size_type = typename rowmap_t::value_type;
  using lno_t     = typename cols_t::value_type;
  using scalar_t  = typename values_t::value_type;
  const std::string name(filename_);
  std::vector<size_type> xadj; std::vector<lno_t> adj; std::vector<scalar_t> ew;
  lno_t nr = 0, nc = 0;
  bool have_nc = false;
  if (kkamd_endswith(name, ".mtx") || kkamd_endswith(name, ".mm")) {
    read_mtx<lno_t, size_type, scalar_t>(filename_, &nr, &nc, xadj, adj, ew, false, false, false);     // :1262
    have_nc = true;
  } else if (kkamd_endswith(name, ".bin")) {                                                            // :681-694
    std::ifstream in(filename_, std::ios::in | std::ios::binary);
    if (!in.is_open()) throw std::runtime_error("File cannot be opened\n");
    size_type ne = 0;
    in.read((char*)&nr, sizeof(lno_t)); in.read((char*)&ne, sizeof(size_type));
    xadj.resize((size_t)nr + 1); adj.resize((size_t)ne); ew.resize((size_t)ne);
    in.read((char*)xadj.data(), sizeof(size_type) * xadj.size());
    in.read((char*)adj.data(), sizeof(lno_t) * adj.size());
    in.read((char*)ew.data(), sizeof(scalar_t) * ew.size());
    if (!in) throw std::runtime_error("truncated .bin file");
  } else if (kkamd_endswith(name, ".crs")) {                                                            // :718-738
    std::ifstream in(filename_);
    if (!in.is_open()) throw std::runtime_error("File cannot be opened\n");
    size_type ne = 0;
    in >> nr >> ne;
    xadj.resize((size_t)nr + 1); adj.resize((size_t)ne); ew.assign((size_t)ne, scalar_t(0));
    for (auto& v : xadj) in >> v;
    for (auto& v : adj) in >> v;
    for (auto& v : ew) { double w; if (in >> w) v = (scalar_t)w; }
  } else {
    throw std::runtime_error("Reader is not available\n");
  }
  if (!have_nc) {                               // .crs and .bin do not store the column count (:1281-1284)
    nc = 0;
    for (lno_t c : adj) nc = std::max(nc, (lno_t)(c + 1));
  }
  rowmap_t rowmap("rowmap_view", (size_t)nr + 1);
  cols_t columns("colsmap_view", adj.size());
  values_t values("values_view", adj.size());
  Kokkos::deep_copy(rowmap, Kokkos::View<size_type*, Kokkos::HostSpace>(xadj.data(), xadj.size()));
  if (!adj.empty()) {
    Kokkos::deep_copy(columns, Kokkos::View<lno_t*, Kokkos::HostSpace>(adj.data(), adj.size()));
    Kokkos::deep_copy(values, Kokkos::View<scalar_t*, Kokkos::HostSpace>(ew.data(), ew.size()));
  }
  return crsMat_t("CrsMatrix", nr, nc, adj.size(), values, rowmap, columns);
}

// .mtx / .mm (coordinate real general, 17 digits), or .bin / .crs for square matrices (:741-783)
template <typename crs_matrix_t>
void write_kokkos_crst_matrix(crs_matrix_t A, const char* filename) {
  using size_type = typename crs_matrix_t::non_const_size_type;
  using lno_t     = typename crs_matrix_t::non_const_ordinal_type;
  using scalar_t  = typename crs_matrix_t::non_const_value_type;
  const std::string name(filename);
  auto rm = Kokkos::create_mirror_view(A.graph.row_map); Kokkos::deep_copy(rm, A.graph.row_map);
  auto en = Kokkos::create_mirror_view(A.graph.entries); Kokkos::deep_copy(en, A.graph.entries);
  auto va = Kokkos::create_mirror_view(A.values);        Kokkos::deep_copy(va, A.values);
  const lno_t nr = A.numRows();
  const size_type ne = (size_type)A.nnz();
  if (kkamd_endswith(name, ".mtx") || kkamd_endswith(name, ".mm")) {
    std::ofstream out(filename);
    out << "%%MatrixMarket matrix coordinate real general\n" << nr << " " << A.numCols() << " " << ne << '\n';
    out << std::setprecision(17) << std::scientific;
    for (lno_t i = 0; i < nr; ++i)
      for (size_type j = rm(i); j < rm(i + 1); ++j) out << i + 1 << " " << en(j) + 1 << " " << va(j) << '\n';
    return;
  }
  if (A.numRows() != A.numCols())
    throw std::runtime_error("For formats other than MatrixMarket (suffix .mm or .mtx),\nwrite_kokkos_crst_matrix only supports square matrices");
  if (kkamd_endswith(name, ".bin")) {
    std::ofstream out(filename, std::ios::out | std::ios::binary);
    out.write((const char*)&nr, sizeof(lno_t)); out.write((const char*)&ne, sizeof(size_type));
    out.write((const char*)rm.data(), sizeof(size_type) * ((size_t)nr + 1));
    out.write((const char*)en.data(), sizeof(lno_t) * (size_t)ne);
    out.write((const char*)va.data(), sizeof(scalar_t) * (size_t)ne);
  } else if (kkamd_endswith(name, ".crs")) {
    std::ofstream out(filename);
    out << nr << " " << ne << "\n";
    for (lno_t i = 0; i <= nr; ++i) out << rm(i) << " ";
    out << "\n";
    for (lno_t i = 0; i < nr; ++i) {
      for (size_type j = rm(i); j < rm(i + 1); ++j) out << en(j) << " ";
      out << "\n";
    }
    out << std::setprecision(17) << std::scientific;      // the reference's writer stops at the graph; its reader reads values
    for (size_type j = 0; j < ne; ++j) out << va(j) << " ";
    out << "\n";
  } else {
    throw std::runtime_error(std::string("write_kokkos_crst_matrix: File extension on ") + filename + " does not correspond to a known format");
  }
}

}  // namespace Impl
}  // namespace KokkosSparse
