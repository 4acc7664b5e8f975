#include "checkpoint.h"
#include <dlfcn.h>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <stdexcept>

namespace helfem {
namespace {

// the part of the HDF5 C API the reference's Checkpoint class uses (hid_t is 64 bits from HDF5 1.10 on)
typedef int64_t hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;
typedef int htri_t;

struct H5 {
  void *lib = nullptr;
  std::string name, why;
  herr_t (*open)() = nullptr;
  hid_t (*Fcreate)(const char *, unsigned, hid_t, hid_t) = nullptr;
  hid_t (*Fopen)(const char *, unsigned, hid_t) = nullptr;
  herr_t (*Fclose)(hid_t) = nullptr;
  hid_t (*Screate_simple)(int, const hsize_t *, const hsize_t *) = nullptr;
  hid_t (*Screate)(int) = nullptr;
  herr_t (*Sclose)(hid_t) = nullptr;
  int (*Sget_simple_extent_ndims)(hid_t) = nullptr;
  int (*Sget_simple_extent_dims)(hid_t, hsize_t *, hsize_t *) = nullptr;
  hid_t (*Dcreate2)(hid_t, const char *, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
  hid_t (*Dopen2)(hid_t, const char *, hid_t) = nullptr;
  herr_t (*Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void *) = nullptr;
  herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void *) = nullptr;
  hid_t (*Dget_space)(hid_t) = nullptr;
  herr_t (*Dclose)(hid_t) = nullptr;
  hid_t (*Tcopy)(hid_t) = nullptr;
  herr_t (*Tclose)(hid_t) = nullptr;
  htri_t (*Lexists)(hid_t, const char *, hid_t) = nullptr;
  herr_t (*Ldelete)(hid_t, const char *, hid_t) = nullptr;
  herr_t (*Eset_auto2)(hid_t, void *, void *) = nullptr;
  herr_t (*Eget_auto2)(hid_t, void **, void **) = nullptr;
  herr_t (*get_libversion)(unsigned *, unsigned *, unsigned *) = nullptr;
  hid_t native_double = -1, native_int = -1, native_hbool = -1;
};

H5 &h5() {
  static H5 api;
  static std::once_flag once;
  std::call_once(once, []() {
    std::vector<std::string> cand;
    if (const char *e = getenv("HELFEM_HDF5_LIB")) cand.push_back(e);
    for (const char *n : {"libhdf5.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5.so.310", "libhdf5_serial.so", "libhdf5_serial.so.103",
                          "/opt/conda/lib/libhdf5.so"})
      cand.push_back(n);
    for (const std::string &n : cand) {
      api.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (api.lib) {
        api.name = n;
        break;
      }
      const char *de = dlerror();  // reading it clears it
      api.why += n + ": " + (de ? de : "?") + "; ";
    }
    if (!api.lib) return;
    bool ok = true;
    auto sym = [&](const char *s) {
      void *p = dlsym(api.lib, s);
      if (!p) {
        ok = false;
        api.why += std::string("missing symbol ") + s + "; ";
      }
      return p;
    };
#define HFG_H5(field, symbol) api.field = reinterpret_cast<decltype(api.field)>(sym(symbol))
    HFG_H5(open, "H5open");
    HFG_H5(Fcreate, "H5Fcreate");
    HFG_H5(Fopen, "H5Fopen");
    HFG_H5(Fclose, "H5Fclose");
    HFG_H5(Screate_simple, "H5Screate_simple");
    HFG_H5(Screate, "H5Screate");
    HFG_H5(Sclose, "H5Sclose");
    HFG_H5(Sget_simple_extent_ndims, "H5Sget_simple_extent_ndims");
    HFG_H5(Sget_simple_extent_dims, "H5Sget_simple_extent_dims");
    HFG_H5(Dcreate2, "H5Dcreate2");
    HFG_H5(Dopen2, "H5Dopen2");
    HFG_H5(Dwrite, "H5Dwrite");
    HFG_H5(Dread, "H5Dread");
    HFG_H5(Dget_space, "H5Dget_space");
    HFG_H5(Dclose, "H5Dclose");
    HFG_H5(Tcopy, "H5Tcopy");
    HFG_H5(Tclose, "H5Tclose");
    HFG_H5(Lexists, "H5Lexists");
    HFG_H5(Ldelete, "H5Ldelete");
    HFG_H5(Eset_auto2, "H5Eset_auto2");
    HFG_H5(Eget_auto2, "H5Eget_auto2");
    HFG_H5(get_libversion, "H5get_libversion");
#undef HFG_H5
    // the prototypes above assume the 64-bit hid_t of HDF5 >= 1.10; an older library would take and return handles of the
    // wrong width.  H5get_libversion needs no initialised library.
    if (ok) {
      unsigned maj = 0, min = 0, rel = 0;
      if (api.get_libversion(&maj, &min, &rel) < 0 || maj < 1 || (maj == 1 && min < 10)) {
        ok = false;
        api.why += api.name + ": HDF5 " + std::to_string(maj) + "." + std::to_string(min) + "." + std::to_string(rel) +
                   " is older than 1.10 (32-bit handles); ";
      }
    }
    if (ok && api.open() >= 0) {
      // the H5T_NATIVE_* macros are globals that H5open() fills in
      hid_t *d = (hid_t *)sym("H5T_NATIVE_DOUBLE_g"), *i = (hid_t *)sym("H5T_NATIVE_INT_g"), *b = (hid_t *)sym("H5T_NATIVE_HBOOL_g");
      if (d && i && b) {
        api.native_double = *d;
        api.native_int = *i;
        api.native_hbool = *b;
      } else
        ok = false;
    } else
      ok = false;
    if (!ok) {
      dlclose(api.lib);
      api.lib = nullptr;
    }
  });
  return api;
}

const H5 &need() {
  H5 &a = h5();
  if (!a.lib)
    throw std::runtime_error("Checkpoint: no usable HDF5 library in this process (" + a.why +
                             "set HELFEM_HDF5_LIB to a libhdf5.so, or run with --save \"\")\n");
  return a;
}

// Calls that may fail as a matter of course (a file or an entry that is not there) are made with the library's automatic
// error printing switched off and the caller's handler restored afterwards: the process may host other HDF5 users
// (h5py in the same interpreter), whose error reporting is theirs.  Failures come back as exceptions here.
struct Quiet {
  const H5 &a;
  void *func = nullptr, *data = nullptr;
  bool saved = false;
  explicit Quiet(const H5 &api) : a(api) {
    saved = a.Eget_auto2(0, &func, &data) >= 0;
    if (saved) a.Eset_auto2(0, nullptr, nullptr);
  }
  ~Quiet() {
    if (saved) a.Eset_auto2(0, func, data);
  }
};

constexpr unsigned F_ACC_RDONLY = 0u, F_ACC_TRUNC = 2u;
constexpr int S_SCALAR = 0;

struct Dataset {  // RAII: an open dataset and its dataspace
  const H5 &a;
  hid_t d = -1, s = -1;
  Dataset(const H5 &api, hid_t file, const std::string &name) : a(api) {
    {
      Quiet q(a);
      d = a.Dopen2(file, name.c_str(), 0);
    }
    if (d < 0) throw std::runtime_error("The entry " + name + " does not exist in the checkpoint file!\n");
    s = a.Dget_space(d);
  }
  ~Dataset() {
    if (s >= 0) a.Sclose(s);
    if (d >= 0) a.Dclose(d);
  }
  std::vector<hsize_t> dims() const {
    int nd = a.Sget_simple_extent_ndims(s);
    std::vector<hsize_t> dm(nd > 0 ? nd : 0);
    if (nd > 0) a.Sget_simple_extent_dims(s, dm.data(), nullptr);
    return dm;
  }
};
}  // namespace

bool hdf5_available(std::string *why) {
  H5 &a = h5();
  if (why) *why = a.lib ? a.name : a.why;
  return a.lib != nullptr;
}

Checkpoint::Checkpoint(const std::string &fname, bool write) : write_(write) {
  const H5 &a = need();
  {
    Quiet q(a);
    file_ = write ? a.Fcreate(fname.c_str(), F_ACC_TRUNC, 0, 0) : a.Fopen(fname.c_str(), F_ACC_RDONLY, 0);
  }
  if (file_ < 0) throw std::runtime_error("Trying to open nonexistent or unwritable checkpoint file \"" + fname + "\"!\n");
}

Checkpoint::~Checkpoint() {
  if (file_ >= 0) h5().Fclose(file_);
}

bool Checkpoint::exist(const std::string &name) const {
  Quiet q(need());
  return need().Lexists(file_, name.c_str(), 0) > 0;
}

void Checkpoint::remove(const std::string &name) {
  if (!write_) throw std::runtime_error("Cannot write to checkpoint file that was opened for reading only!\n");
  if (exist(name)) need().Ldelete(file_, name.c_str(), 0);
}

static void write_2d(const H5 &a, hid_t file, const std::string &name, hid_t type, hsize_t d0, hsize_t d1, const void *data) {
  hsize_t dims[2] = {d0, d1};
  hid_t space = a.Screate_simple(2, dims, nullptr);
  hid_t dt = a.Tcopy(type);
  hid_t ds = a.Dcreate2(file, name.c_str(), dt, space, 0, 0, 0);
  herr_t rc = ds >= 0 ? a.Dwrite(ds, dt, 0, 0, 0, data) : -1;
  if (ds >= 0) a.Dclose(ds);
  a.Tclose(dt);
  a.Sclose(space);
  if (rc < 0) throw std::runtime_error("Checkpoint: writing " + name + " failed\n");
}

void Checkpoint::write(const std::string &name, const Mat &m) {
  remove(name);
  // dims[1] = n_rows, dims[0] = n_cols: column-major memory written as a row-major n_cols x n_rows dataset
  static const double none = 0.0;
  write_2d(need(), file_, name, need().native_double, m.n_cols, m.n_rows, m.n_elem() ? (const void *)m.memptr() : (const void *)&none);
}

void Checkpoint::write(const std::string &name, const Vec &v) {
  Mat m(v.size(), 1);
  m.d = v;
  write(name, m);
}

void Checkpoint::write(const std::string &name, const IVec &v) {
  remove(name);
  static const int none = 0;
  write_2d(need(), file_, name, need().native_int, v.size(), 1, v.size() ? (const void *)v.data() : (const void *)&none);
}

static void write_scalar(const H5 &a, hid_t file, const std::string &name, hid_t type, const void *val) {
  hid_t space = a.Screate(S_SCALAR);
  hid_t dt = a.Tcopy(type);
  hid_t ds = a.Dcreate2(file, name.c_str(), dt, space, 0, 0, 0);
  herr_t rc = ds >= 0 ? a.Dwrite(ds, dt, 0, 0, 0, val) : -1;
  if (ds >= 0) a.Dclose(ds);
  a.Tclose(dt);
  a.Sclose(space);
  if (rc < 0) throw std::runtime_error("Checkpoint: writing " + name + " failed\n");
}

void Checkpoint::write(const std::string &name, double v) {
  remove(name);
  write_scalar(need(), file_, name, need().native_double, &v);
}
void Checkpoint::write(const std::string &name, int v) {
  remove(name);
  write_scalar(need(), file_, name, need().native_int, &v);
}
void Checkpoint::write_bool(const std::string &name, bool v) {
  remove(name);
  unsigned int buf = v ? 1u : 0u;  // hbool_t is bool or unsigned depending on the library's configuration: both read this right
  write_scalar(need(), file_, name, need().native_hbool, &buf);
}

void Checkpoint::write(const diatomic::TwoDBasis &basis) {
  write("HelFEM_ID", 2);
  write("Z1", basis.Z1);
  write("Z2", basis.Z2);
  write("Rhalf", basis.Rhalf);
  write("bval", basis.fem.bval);
  write("n_quad", basis.nquad());
  write("poly_id", basis.primbas);
  write("poly_nnodes", basis.nnodes);
  write("lval", basis.lval);
  write("mval", basis.mval);
}

void Checkpoint::write(const atomic::TwoDBasis &basis) {
  write("HelFEM_ID", 1);
  write("Z", basis.Z);
  write("Zl", 0);
  write("Zr", 0);
  write("Rhalf", 0.0);
  write("bval", basis.fem.bval);
  write("finitenuc", 0);
  write("Rrms", 0.0);
  write("n_quad", (int)basis.xq.size());
  write("poly_id", 4);
  write("poly_nnodes", basis.nnodes);
  write_bool("zeroder", false);
  write("taylor_order", -1);
  write("lval", basis.lval);
  write("mval", basis.mval);
}

std::vector<long long> Checkpoint::dims(const std::string &name) const {
  Dataset d(need(), file_, name);
  std::vector<long long> out;
  for (hsize_t v : d.dims()) out.push_back((long long)v);
  return out;
}

void Checkpoint::read(const std::string &name, Mat &m) const {
  const H5 &a = need();
  Dataset d(a, file_, name);
  std::vector<hsize_t> dm = d.dims();
  if (dm.size() != 2) throw std::runtime_error("Error - " + name + " should have dimension 2.\n");
  m.zeros((size_t)dm[1], (size_t)dm[0]);
  if (m.n_elem() && a.Dread(d.d, a.native_double, 0, 0, 0, m.memptr()) < 0) throw std::runtime_error("Checkpoint: reading " + name + " failed\n");
}

void Checkpoint::read(const std::string &name, Vec &v) const {
  Mat m;
  read(name, m);
  v = m.d;
}

void Checkpoint::read(const std::string &name, IVec &v) const {
  const H5 &a = need();
  Dataset d(a, file_, name);
  std::vector<hsize_t> dm = d.dims();
  if (dm.size() != 2) throw std::runtime_error("Error - " + name + " should have dimension 2.\n");
  v.assign((size_t)(dm[0] * dm[1]), 0);
  if (!v.empty() && a.Dread(d.d, a.native_int, 0, 0, 0, v.data()) < 0) throw std::runtime_error("Checkpoint: reading " + name + " failed\n");
}

void Checkpoint::read(const std::string &name, double &v) const {
  const H5 &a = need();
  Dataset d(a, file_, name);
  if (a.Dread(d.d, a.native_double, 0, 0, 0, &v) < 0) throw std::runtime_error("Checkpoint: reading " + name + " failed\n");
}

void Checkpoint::read(const std::string &name, int &v) const {
  const H5 &a = need();
  Dataset d(a, file_, name);
  if (a.Dread(d.d, a.native_int, 0, 0, 0, &v) < 0) throw std::runtime_error("Checkpoint: reading " + name + " failed\n");
}

diatomic::TwoDBasis Checkpoint::read_diatomic_basis(int lpad) const {
  int id = 0;
  read("HelFEM_ID", id);
  if (id != 2) throw std::logic_error("Checkpoint does not correspond to a diatomic calculation!\n");
  int Z1, Z2, nq, pid, nn;
  double Rh;
  Vec bval;
  IVec lval, mval;
  read("Z1", Z1);
  read("Z2", Z2);
  read("Rhalf", Rh);
  read("bval", bval);
  read("n_quad", nq);
  read("poly_id", pid);
  read("poly_nnodes", nn);
  read("lval", lval);
  read("mval", mval);
  if (pid != 4) throw std::logic_error("Only the LIP primitive basis (poly_id 4) is supported by this build.\n");
  return diatomic::TwoDBasis(Z1, Z2, Rh, nn, nq, bval, lval, mval, lpad);
}

atomic::TwoDBasis Checkpoint::read_atomic_basis() const {
  int id = 0;
  read("HelFEM_ID", id);
  if (id != 1) throw std::logic_error("Checkpoint does not correspond to an atomic calculation!\n");
  int Z, nq, pid, nn, finitenuc = 0;
  Vec bval;
  IVec lval, mval;
  read("Z", Z);
  read("bval", bval);
  read("n_quad", nq);
  read("poly_id", pid);
  read("poly_nnodes", nn);
  read("lval", lval);
  read("mval", mval);
  if (exist("finitenuc")) read("finitenuc", finitenuc);
  if (pid != 4) throw std::logic_error("Only the LIP primitive basis (poly_id 4) is supported by this build.\n");
  if (finitenuc != 0) throw std::logic_error("Finite nuclear models are not supported by this build.\n");
  return atomic::TwoDBasis(Z, nn, nq, bval, lval, mval);
}

}  // namespace helfem
