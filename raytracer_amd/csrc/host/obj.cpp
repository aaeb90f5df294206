// obj.cpp — Wavefront OBJ reader matching what the reference gets from
// tobj 4.0.3 (Cargo.lock) with GPU_LOAD_OPTIONS and single_index = false
// (src/mesh.rs:122-127): triangulate = true, points and lines ignored, only
// the FIRST model of the file is used (`.0[0].mesh`).
//
// tobj's source is not vendored in the reference, so these rules restate its
// published behaviour:
//   - `v x y z`, `vn x y z`, `vt u v` are collected file-wide;
//   - `f` corners are `v`, `v/vt`, `v//vn` or `v/vt/vn`; indices are 1-based,
//     negative values count back from the current end of the respective list;
//   - an n-gon is fan-triangulated as (v0, v[i], v[i+1]);
//   - a new model starts at an `o` or `g` statement once the current one has
//     faces;
//   - with single_index = false each model gets its own position array holding
//     the vertices it uses in order of first use (same, separately, for
//     normals), and `indices` refer to that array;
//   - floats are parsed correctly rounded (strtof == Rust's f32::from_str).
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "host.hpp"

namespace mirth {
namespace {

struct Corner {
  long v = 0, vt = 0, vn = 0;  // 0 = missing; already resolved to 0-based + 1
};

const char *skip_ws(const char *p, const char *e) {
  while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
  return p;
}

bool parse_float(const char *&p, const char *e, float &out) {
  p = skip_ws(p, e);
  if (p >= e || *p == '\n') return false;
  char buf[64];
  size_t n = 0;
  while (p < e && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n' && n + 1 < sizeof(buf)) buf[n++] = *p++;
  buf[n] = 0;
  char *endp = nullptr;
  out = strtof(buf, &endp);
  return endp != buf;
}

// A decimal integer with an optional sign, read without looking at or past `e`
// (the text need not end in a NUL or a newline).  False when no digit is there.
bool parse_long(const char *&p, const char *e, long &out) {
  const char *c = p;
  bool neg = false;
  if (c < e && (*c == '-' || *c == '+')) neg = *c++ == '-';
  if (c >= e || *c < '0' || *c > '9') return false;
  long v = 0;
  while (c < e && *c >= '0' && *c <= '9') {
    if (v > (0x7FFFFFFFL - 9) / 10) return false;  // no mesh has 2^31 vertices
    v = v * 10 + (*c++ - '0');
  }
  out = neg ? -v : v;
  p = c;
  return true;
}

// Resolve a 1-based / negative OBJ index against a list of `count` items.
bool resolve(long raw, size_t count, long &out) {
  if (raw > 0) {
    out = raw;  // keep 1-based, 0 means missing
    return (size_t)raw <= count;
  }
  if (raw < 0) {
    long idx = (long)count + raw;  // 0-based
    if (idx < 0) return false;
    out = idx + 1;
    return true;
  }
  return false;
}

}  // namespace

bool parse_obj_first_model(const char *text, size_t len, ObjModel &out, std::string &err) {
  std::vector<float> pos, nrm;
  size_t n_tex = 0;
  std::vector<std::vector<Corner>> faces;  // of the first model only
  bool first_model_closed = false;

  const char *p = text, *e = text + len;
  size_t line_no = 0;
  while (p < e) {
    ++line_no;
    const char *eol = (const char *)memchr(p, '\n', (size_t)(e - p));
    if (!eol) eol = e;
    const char *q = skip_ws(p, eol);
    if (q < eol && *q != '#') {
      if (q + 1 < eol && q[0] == 'v' && (q[1] == ' ' || q[1] == '\t')) {
        const char *c = q + 1;
        float f[3];
        for (int i = 0; i < 3; ++i)
          if (!parse_float(c, eol, f[i])) {
            err = "bad vertex at line " + std::to_string(line_no);
            return false;
          }
        pos.insert(pos.end(), f, f + 3);
      } else if (q + 2 < eol && q[0] == 'v' && q[1] == 'n' && (q[2] == ' ' || q[2] == '\t')) {
        const char *c = q + 2;
        float f[3];
        for (int i = 0; i < 3; ++i)
          if (!parse_float(c, eol, f[i])) {
            err = "bad normal at line " + std::to_string(line_no);
            return false;
          }
        nrm.insert(nrm.end(), f, f + 3);
      } else if (q + 2 < eol && q[0] == 'v' && q[1] == 't' && (q[2] == ' ' || q[2] == '\t')) {
        ++n_tex;
      } else if (q + 1 < eol && q[0] == 'f' && (q[1] == ' ' || q[1] == '\t')) {
        if (!first_model_closed) {
          std::vector<Corner> face;
          const char *c = q + 1;
          while (true) {
            c = skip_ws(c, eol);
            if (c >= eol) break;
            Corner k;
            long raw[3] = {0, 0, 0};
            int field = 0;
            while (c < eol && *c != ' ' && *c != '\t' && *c != '\r') {
              if (*c == '/') {
                ++field;
                ++c;
                if (field > 2) break;
                continue;
              }
              long val = 0;
              if (!parse_long(c, eol, val)) {
                err = "bad face index at line " + std::to_string(line_no);
                return false;
              }
              raw[field] = val;
            }
            if (!resolve(raw[0], pos.size() / 3, k.v)) {
              err = "face vertex out of bounds at line " + std::to_string(line_no);
              return false;
            }
            if (raw[1] != 0 && !resolve(raw[1], n_tex, k.vt)) k.vt = 0;
            if (raw[2] != 0 && !resolve(raw[2], nrm.size() / 3, k.vn)) {
              err = "face normal out of bounds at line " + std::to_string(line_no);
              return false;
            }
            face.push_back(k);
          }
          if (face.size() >= 3) faces.push_back(std::move(face));
          // points (1 corner) and lines (2 corners) are ignored
        }
      } else if ((q[0] == 'o' || q[0] == 'g') && (q + 1 >= eol || q[1] == ' ' || q[1] == '\t' || q[1] == '\r')) {
        if (!faces.empty()) first_model_closed = true;
      }
      // everything else (mtllib, usemtl, s, ...) does not affect geometry
    }
    p = eol + 1;
  }

  if (faces.empty()) {
    err = "no faces in first model";
    return false;
  }

  out = ObjModel();
  std::unordered_map<long, uint32_t> vmap, nmap;
  const bool have_normals = !nrm.empty();
  auto add_corner = [&](const Corner &k) -> bool {
    auto it = vmap.find(k.v);
    if (it != vmap.end()) {
      out.indices.push_back(it->second);
    } else {
      size_t src = (size_t)(k.v - 1) * 3;
      out.positions.push_back(pos[src]);
      out.positions.push_back(pos[src + 1]);
      out.positions.push_back(pos[src + 2]);
      uint32_t next = (uint32_t)vmap.size();
      out.indices.push_back(next);
      vmap.emplace(k.v, next);
    }
    if (have_normals) {
      if (k.vn == 0) return false;  // tobj: a file with normals needs one per corner
      if (nmap.find(k.vn) == nmap.end()) {
        size_t src = (size_t)(k.vn - 1) * 3;
        out.normals.push_back(nrm[src]);
        out.normals.push_back(nrm[src + 1]);
        out.normals.push_back(nrm[src + 2]);
        uint32_t next = (uint32_t)nmap.size();
        nmap.emplace(k.vn, next);
      }
    }
    return true;
  };
  for (const auto &f : faces) {
    for (size_t i = 1; i + 1 < f.size(); ++i) {
      if (!add_corner(f[0]) || !add_corner(f[i]) || !add_corner(f[i + 1])) {
        err = "face corner without a normal index in a file that has normals";
        return false;
      }
    }
  }
  return true;
}

bool load_obj_first_model(const std::string &path, ObjModel &out, std::string &err) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) {
    err = "cannot open " + path + ": " + strerror(errno);
    return false;
  }
  std::vector<char> buf;
  char chunk[1 << 16];
  size_t n;
  while ((n = fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + n);
  fclose(f);
  return parse_obj_first_model(buf.data(), buf.size(), out, err);
}

}  // namespace mirth
