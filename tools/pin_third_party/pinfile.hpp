// pinfile.hpp -- the array container of the third-party pin harness (see pinfile.py for the format).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace pin {

enum Code : uint32_t { U8 = 0, I32 = 1, F32 = 2, F64 = 3, U16 = 4, I64 = 5 };
inline size_t code_size(uint32_t c) { static const size_t s[6] = {1, 4, 4, 8, 2, 8}; return c < 6 ? s[c] : 0; }

struct Array {
  uint32_t code = U8;
  std::vector<uint64_t> dims;
  std::vector<unsigned char> bytes;
  size_t count() const { size_t n = 1; for (uint64_t d : dims) n *= (size_t)d; return n; }
  template <typename T> const T* as() const { return reinterpret_cast<const T*>(bytes.data()); }
};

inline std::map<std::string, Array> read(const std::string& path) {
  std::map<std::string, Array> out;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  char magic[8];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "RGBDPIN1", 8) != 0) { fclose(f); throw std::runtime_error(path + " is not a pin file"); }
  for (;;) {
    uint32_t nl;
    if (fread(&nl, 4, 1, f) != 1) break;
    std::string name(nl, '\0');
    Array a;
    uint32_t ndim;
    if (fread(&name[0], 1, nl, f) != nl || fread(&a.code, 4, 1, f) != 1 || fread(&ndim, 4, 1, f) != 1) { fclose(f); throw std::runtime_error("truncated record"); }
    a.dims.resize(ndim);
    if (ndim && fread(a.dims.data(), 8, ndim, f) != ndim) { fclose(f); throw std::runtime_error("truncated dims"); }
    a.bytes.resize(a.count() * code_size(a.code));
    if (!a.bytes.empty() && fread(a.bytes.data(), 1, a.bytes.size(), f) != a.bytes.size()) { fclose(f); throw std::runtime_error("truncated data"); }
    out[name] = a;
  }
  fclose(f);
  return out;
}

class Writer {
 public:
  explicit Writer(const std::string& path) : f_(fopen(path.c_str(), "wb")) {
    if (!f_) throw std::runtime_error("cannot create " + path);
    fwrite("RGBDPIN1", 1, 8, f_);
  }
  ~Writer() { if (f_) fclose(f_); }
  void put(const std::string& name, uint32_t code, const std::vector<uint64_t>& dims, const void* data) {
    const uint32_t nl = (uint32_t)name.size(), ndim = (uint32_t)dims.size();
    size_t n = 1;
    for (uint64_t d : dims) n *= (size_t)d;
    fwrite(&nl, 4, 1, f_); fwrite(name.data(), 1, nl, f_); fwrite(&code, 4, 1, f_); fwrite(&ndim, 4, 1, f_);
    if (ndim) fwrite(dims.data(), 8, ndim, f_);
    if (n) fwrite(data, code_size(code), n, f_);
  }
 private:
  FILE* f_;
};

}  // namespace pin
