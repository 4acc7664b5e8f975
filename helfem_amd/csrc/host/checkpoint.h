// HDF5 checkpoint files in the reference's format (/root/reference/src/general/checkpoint.cpp):
//   arma::mat / arma::vec   -> 2-D dataset of native doubles with SWAPPED dimensions, dims = {n_cols, n_rows}   (:117-144)
//   arma::ivec / arma::imat -> 2-D dataset of native ints, dims = {n_rows, n_cols}                                (:220-257)
//   double / int / bool     -> scalar datasets (native double / int / hbool)                                      (:627, :701, :841)
//   basis                   -> the constructor arguments: HelFEM_ID (1 atomic, 2 diatomic), charges, Rhalf, bval, n_quad,
//                              poly_id, poly_nnodes, lval, mval (+ finitenuc, Rrms, zeroder, taylor_order for atoms)  (:477-507, :560-584)
// so that the reference's tools (diatomic_dgrid, diatomic_cpl, ...) can read what this code writes and vice versa.
//
// libhdf5 is bound at RUN time (dlopen): the product neither needs HDF5 headers to build nor the library to run without
// checkpoints.  When a checkpoint is asked for and no libhdf5 can be loaded, opening throws (nothing is skipped silently).
// The library is searched as $HELFEM_HDF5_LIB, libhdf5.so, libhdf5.so.103, libhdf5.so.200, libhdf5_serial.so,
// /opt/conda/lib/libhdf5.so.
#pragma once
#include "atomic_basis.h"
#include "diatomic_basis.h"
#include "linalg.h"
#include <string>

namespace helfem {

class Checkpoint {
 public:
  /// write = true truncates / creates the file, write = false opens it read-only
  Checkpoint(const std::string &fname, bool write);
  ~Checkpoint();
  Checkpoint(const Checkpoint &) = delete;
  Checkpoint &operator=(const Checkpoint &) = delete;

  bool exist(const std::string &name) const;
  void write(const std::string &name, const Mat &m);
  void write(const std::string &name, const Vec &v);   // arma::vec: an n x 1 matrix, dims {1, n}
  void write(const std::string &name, const IVec &v);  // arma::ivec: dims {n, 1}
  void write(const std::string &name, double v);
  void write(const std::string &name, int v);
  void write_bool(const std::string &name, bool v);
  void write(const diatomic::TwoDBasis &basis);
  void write(const atomic::TwoDBasis &basis);

  void read(const std::string &name, Mat &m) const;
  void read(const std::string &name, Vec &v) const;
  void read(const std::string &name, IVec &v) const;
  void read(const std::string &name, double &v) const;
  void read(const std::string &name, int &v) const;
  /// the constructor arguments of a diatomic basis (Checkpoint::read(diatomic::basis::TwoDBasis &), checkpoint.cpp:587-625)
  diatomic::TwoDBasis read_diatomic_basis(int lpad) const;
  /// the constructor arguments of an atomic basis (Checkpoint::read(atomic::basis::TwoDBasis &), checkpoint.cpp:510-558);
  /// point nucleus, LIP primitives
  atomic::TwoDBasis read_atomic_basis() const;

  /// dataset shape as stored (HDF5 order); empty for scalars
  std::vector<long long> dims(const std::string &name) const;

 private:
  void remove(const std::string &name);
  long long file_ = -1;
  bool write_ = false;
};

/// true when a libhdf5 can be loaded in this process
bool hdf5_available(std::string *why = nullptr);

}  // namespace helfem
