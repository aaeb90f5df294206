// gltf.cpp — a glTF 2.0 loader for the host scene model.
//
// The reference sketches one and leaves it commented out (src/scenes.rs:332-399: per primitive
// positions, indices and normals into a mesh, base_color_factor into a material, an emissive
// factor into a light material, one instance per primitive with an identity transform, node
// matrices computed and dropped) and carries texture ids in Material that nothing fills
// (src/material.rs:15-18).  This is that loader finished, for the scene model as it stands now
// (one PBR material struct, a texture store):
//   - .gltf (JSON, buffers as files beside it or base64 data URIs) and .glb (binary container);
//   - every TRIANGLES primitive of every mesh a node of the default scene refers to becomes one
//     mesh of the MeshServer, shared by all nodes that refer to it.  Positions are taken as they
//     are (no normalisation: that is the OBJ path's, src/mesh.rs:308-355), normals from the file
//     or computed as for an OBJ without normals, TEXCOORD_0 into the vertex uvs;
//   - a node's world matrix T·R·S becomes the instance's Transform (scale, Euler rotation with
//     R = Rx·Ry·Rz, translation) when it is one — no shear; otherwise the matrix is baked into a
//     private copy of the mesh and the instance gets the identity;
//   - materials: baseColorFactor, metallicFactor, roughnessFactor, emissiveFactor (times
//     KHR_materials_emissive_strength), KHR_materials_ior, KHR_materials_transmission,
//     baseColorTexture and emissiveTexture (PNG images, 8 or 16 bits per channel, not
//     interlaced; sampler magFilter NEAREST -> nearest, anything else bilinear; repeat
//     addressing whatever the sampler says).  A primitive without a material gets glTF's
//     default (white, metallic 1, roughness 1);
//   - skipped, and counted in MirthGltfInfo: primitives of other modes, JPEG and other images,
//     sparse accessors, cameras, lights, skins, animations, morph targets.
// Nothing here has a counterpart in the reference to be bit-identical with; tests/test_gltf.py
// checks it against files the test writes itself.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

#include "host.hpp"
#include "mirt_host.h"

namespace mirth {
namespace {

// ------------------------------------------------------------------ JSON
struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;

  const Json *get(const char *key) const {
    if (kind != Obj) return nullptr;
    for (const auto &kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  bool has(const char *key) const { return get(key) != nullptr; }
  double number(const char *key, double dflt) const {
    const Json *j = get(key);
    return j && j->kind == Num ? j->num : dflt;
  }
  // a non-negative integer index, or -1
  long index(const char *key) const {
    const Json *j = get(key);
    if (!j || j->kind != Num || j->num < 0.0 || j->num > 4.0e9 || j->num != std::floor(j->num)) return -1;
    return (long)j->num;
  }
  size_t size() const { return kind == Arr ? arr.size() : 0; }
  // this value itself as a non-negative integer index, or -1 (the same test as index(key))
  long as_index() const {
    if (kind != Num || !(num >= 0.0) || num > 4.0e9 || num != std::floor(num)) return -1;
    return (long)num;
  }
};

struct JsonParser {
  const char *p, *end;
  std::string err;
  int depth = 0;

  void ws() {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
  }
  bool fail(const char *m) {
    if (err.empty()) err = m;
    return false;
  }
  bool value(Json &out) {
    if (++depth > 200) return fail("JSON nested too deeply");
    ws();
    if (p >= end) return fail("JSON ends early");
    bool ok = false;
    const char c = *p;
    if (c == '{') ok = object(out);
    else if (c == '[') ok = array(out);
    else if (c == '"') { out.kind = Json::Str; ok = string(out.str); }
    else if (c == 't' || c == 'f' || c == 'n') ok = literal(out);
    else ok = number(out);
    --depth;
    return ok;
  }
  bool literal(Json &out) {
    auto take = [&](const char *w) {
      const size_t n = strlen(w);
      if ((size_t)(end - p) >= n && memcmp(p, w, n) == 0) { p += n; return true; }
      return false;
    };
    if (take("true")) { out.kind = Json::Bool; out.b = true; return true; }
    if (take("false")) { out.kind = Json::Bool; out.b = false; return true; }
    if (take("null")) { out.kind = Json::Null; return true; }
    return fail("bad JSON literal");
  }
  bool number(Json &out) {
    const char *s = p;
    if (p < end && (*p == '-' || *p == '+')) ++p;
    bool digits = false;
    while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) {
      digits = digits || (*p >= '0' && *p <= '9');
      ++p;
    }
    if (!digits || p - s > 63) return fail("bad JSON number");
    char buf[64];
    memcpy(buf, s, (size_t)(p - s));
    buf[p - s] = 0;
    char *e = nullptr;
    out.num = strtod(buf, &e);
    if (e == buf || *e != 0) return fail("bad JSON number");
    out.kind = Json::Num;
    return true;
  }
  bool string(std::string &out) {
    ++p;  // opening quote
    out.clear();
    while (p < end && *p != '"') {
      unsigned char c = (unsigned char)*p++;
      if (c != '\\') { out.push_back((char)c); continue; }
      if (p >= end) return fail("JSON string ends early");
      const char e = *p++;
      switch (e) {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u': {
          if (end - p < 4) return fail("bad \\u escape");
          unsigned v = 0;
          for (int i = 0; i < 4; ++i) {
            const char h = *p++;
            v <<= 4;
            if (h >= '0' && h <= '9') v |= (unsigned)(h - '0');
            else if (h >= 'a' && h <= 'f') v |= (unsigned)(h - 'a' + 10);
            else if (h >= 'A' && h <= 'F') v |= (unsigned)(h - 'A' + 10);
            else return fail("bad \\u escape");
          }
          // UTF-8 of the code unit (surrogate pairs are kept as two units: names only)
          if (v < 0x80) out.push_back((char)v);
          else if (v < 0x800) { out.push_back((char)(0xC0 | (v >> 6))); out.push_back((char)(0x80 | (v & 63))); }
          else { out.push_back((char)(0xE0 | (v >> 12))); out.push_back((char)(0x80 | ((v >> 6) & 63))); out.push_back((char)(0x80 | (v & 63))); }
          break;
        }
        default: return fail("bad JSON escape");
      }
    }
    if (p >= end) return fail("JSON string ends early");
    ++p;  // closing quote
    return true;
  }
  bool array(Json &out) {
    out.kind = Json::Arr;
    ++p;
    ws();
    if (p < end && *p == ']') { ++p; return true; }
    while (true) {
      out.arr.emplace_back();
      if (!value(out.arr.back())) return false;
      ws();
      if (p < end && *p == ',') { ++p; continue; }
      if (p < end && *p == ']') { ++p; return true; }
      return fail("bad JSON array");
    }
  }
  bool object(Json &out) {
    out.kind = Json::Obj;
    ++p;
    ws();
    if (p < end && *p == '}') { ++p; return true; }
    while (true) {
      ws();
      if (p >= end || *p != '"') return fail("bad JSON object key");
      std::string key;
      if (!string(key)) return false;
      ws();
      if (p >= end || *p != ':') return fail("bad JSON object");
      ++p;
      out.obj.emplace_back(std::move(key), Json());
      if (!value(out.obj.back().second)) return false;
      ws();
      if (p < end && *p == ',') { ++p; continue; }
      if (p < end && *p == '}') { ++p; return true; }
      return fail("bad JSON object");
    }
  }
};

// ------------------------------------------------------------------ bytes
typedef std::vector<unsigned char> Bytes;

bool read_file(const std::string &path, Bytes &out, std::string &err) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) { err = "cannot open " + path; return false; }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 0) { fclose(f); err = "cannot size " + path; return false; }
  out.resize((size_t)n);
  const bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  if (!ok) err = "short read of " + path;
  return ok;
}

bool base64_decode(const char *s, size_t n, Bytes &out) {
  out.clear();
  out.reserve(n / 4 * 3);
  uint32_t acc = 0;
  int bits = 0;
  for (size_t i = 0; i < n; ++i) {
    const char c = s[i];
    int v;
    if (c >= 'A' && c <= 'Z') v = c - 'A';
    else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
    else if (c >= '0' && c <= '9') v = c - '0' + 52;
    else if (c == '+' || c == '-') v = 62;
    else if (c == '/' || c == '_') v = 63;
    else if (c == '=' || c == '\n' || c == '\r') continue;
    else return false;
    acc = (acc << 6) | (uint32_t)v;
    bits += 6;
    if (bits >= 8) {
      bits -= 8;
      out.push_back((unsigned char)((acc >> bits) & 0xFFu));
    }
  }
  return true;
}

std::string percent_decode(const std::string &s) {
  std::string o;
  for (size_t i = 0; i < s.size(); ++i) {
    if (s[i] == '%' && i + 2 < s.size() + 0 && isxdigit((unsigned char)s[i + 1]) && isxdigit((unsigned char)s[i + 2])) {
      o.push_back((char)strtol(s.substr(i + 1, 2).c_str(), nullptr, 16));
      i += 2;
    } else {
      o.push_back(s[i]);
    }
  }
  return o;
}

// a buffer's or image's "uri": a data URI or a file relative to the document
bool load_uri(const std::string &uri, const std::string &dir, Bytes &out, std::string &err) {
  if (uri.compare(0, 5, "data:") == 0) {
    const size_t comma = uri.find(',');
    if (comma == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > comma) {
      err = "data URI that is not base64";
      return false;
    }
    if (!base64_decode(uri.data() + comma + 1, uri.size() - comma - 1, out)) {
      err = "bad base64 in a data URI";
      return false;
    }
    return true;
  }
  if (uri.find("://") != std::string::npos) { err = "remote URI " + uri; return false; }
  return read_file(dir + percent_decode(uri), out, err);
}

// ------------------------------------------------------------------ inflate (RFC 1951) and PNG
struct BitReader {
  const unsigned char *p, *end;
  uint32_t acc = 0;
  int n = 0;
  bool over = false;
  uint32_t bits(int k) {
    while (n < k) {
      if (p < end) acc |= (uint32_t)(*p++) << n;
      else over = true;
      n += 8;
    }
    const uint32_t v = acc & ((k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1u));
    acc >>= k;
    n -= k;
    return v;
  }
};

struct Huffman {
  uint16_t count[16], symbol[288];
  void build(const unsigned char *lengths, int n) {
    memset(count, 0, sizeof(count));
    for (int i = 0; i < n; ++i) count[lengths[i]]++;
    count[0] = 0;
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int i = 0; i < n; ++i)
      if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
  }
  int decode(BitReader &br) const {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
      code |= (int)br.bits(1);
      const int c = count[len];
      if (code - c < first) return symbol[index + (code - first)];
      index += c;
      first += c;
      first <<= 1;
      code <<= 1;
      if (br.over) return -1;
    }
    return -1;
  }
};

bool inflate(const unsigned char *src, size_t n, Bytes &out, size_t limit) {
  static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint16_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint16_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  static const unsigned char kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  BitReader br{src, src + n};
  out.clear();
  bool last = false;
  while (!last) {
    last = br.bits(1) != 0;
    const uint32_t type = br.bits(2);
    if (br.over) return false;
    if (type == 0) {
      br.acc = 0;
      br.n = 0;  // to the next byte
      if (br.end - br.p < 4) return false;
      const uint32_t len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
      br.p += 4;
      if ((len ^ 0xFFFFu) != nlen || (size_t)(br.end - br.p) < len || out.size() + len > limit) return false;
      out.insert(out.end(), br.p, br.p + len);
      br.p += len;
      continue;
    }
    if (type == 3) return false;
    Huffman lit, dist;
    unsigned char lengths[320];
    if (type == 1) {
      for (int i = 0; i < 144; ++i) lengths[i] = 8;
      for (int i = 144; i < 256; ++i) lengths[i] = 9;
      for (int i = 256; i < 280; ++i) lengths[i] = 7;
      for (int i = 280; i < 288; ++i) lengths[i] = 8;
      lit.build(lengths, 288);
      for (int i = 0; i < 30; ++i) lengths[i] = 5;
      dist.build(lengths, 30);
    } else {
      const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
      if (nlen > 286 || ndist > 30) return false;
      unsigned char cl[19] = {0};
      for (int i = 0; i < ncode; ++i) cl[kOrder[i]] = (unsigned char)br.bits(3);
      Huffman lencode;
      lencode.build(cl, 19);
      int i = 0;
      while (i < nlen + ndist) {
        const int sym = lencode.decode(br);
        if (sym < 0) return false;
        if (sym < 16) { lengths[i++] = (unsigned char)sym; continue; }
        int rep, val = 0;
        if (sym == 16) {
          if (i == 0) return false;
          val = lengths[i - 1];
          rep = 3 + (int)br.bits(2);
        } else if (sym == 17) {
          rep = 3 + (int)br.bits(3);
        } else {
          rep = 11 + (int)br.bits(7);
        }
        if (i + rep > nlen + ndist) return false;
        while (rep--) lengths[i++] = (unsigned char)val;
      }
      if (lengths[256] == 0) return false;
      lit.build(lengths, nlen);
      dist.build(lengths + nlen, ndist);
    }
    while (true) {
      const int sym = lit.decode(br);
      if (sym < 0 || br.over) return false;
      if (sym < 256) {
        if (out.size() >= limit) return false;
        out.push_back((unsigned char)sym);
        continue;
      }
      if (sym == 256) break;
      if (sym > 285) return false;
      const int len = kLenBase[sym - 257] + (int)br.bits(kLenExtra[sym - 257]);
      const int ds = dist.decode(br);
      if (ds < 0 || ds > 29) return false;
      const size_t d = (size_t)kDistBase[ds] + br.bits(kDistExtra[ds]);
      if (d > out.size() || out.size() + (size_t)len > limit) return false;
      const size_t from = out.size() - d;
      for (int k = 0; k < len; ++k) out.push_back(out[from + (size_t)k]);
    }
  }
  return !br.over;
}

uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// PNG -> RGBA8 texels (one uint32 per texel, red in the low byte), rows top to bottom
bool decode_png(const Bytes &png, uint32_t &width, uint32_t &height, std::vector<uint32_t> &texels, std::string &err) {
  static const unsigned char kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (png.size() < 8 + 25 || memcmp(png.data(), kSig, 8) != 0) { err = "not a PNG"; return false; }
  size_t at = 8;
  uint32_t depth = 0, ctype = 0, interlace = 0;
  Bytes idat;
  unsigned char palette[256][4];
  memset(palette, 255, sizeof(palette));
  uint32_t n_palette = 0;
  bool have_hdr = false, have_trns_key = false;
  uint32_t trns_key[3] = {0, 0, 0};
  while (at + 12 <= png.size()) {
    const uint32_t len = be32(&png[at]);
    const unsigned char *type = &png[at + 4];
    if (len > png.size() - at - 12) { err = "PNG chunk runs past the file"; return false; }
    const unsigned char *data = &png[at + 8];
    if (memcmp(type, "IHDR", 4) == 0 && len >= 13) {
      width = be32(data);
      height = be32(data + 4);
      depth = data[8];
      ctype = data[9];
      interlace = data[12];
      have_hdr = true;
    } else if (memcmp(type, "PLTE", 4) == 0) {
      n_palette = len / 3 < 256 ? len / 3 : 256;
      for (uint32_t i = 0; i < n_palette; ++i) memcpy(palette[i], data + 3 * i, 3);
    } else if (memcmp(type, "tRNS", 4) == 0) {
      if (ctype == 3) {
        for (uint32_t i = 0; i < len && i < 256; ++i) palette[i][3] = data[i];
      } else if (ctype == 0 && len >= 2) {
        have_trns_key = true;
        trns_key[0] = trns_key[1] = trns_key[2] = ((uint32_t)data[0] << 8) | data[1];
      } else if (ctype == 2 && len >= 6) {
        have_trns_key = true;
        for (int k = 0; k < 3; ++k) trns_key[k] = ((uint32_t)data[2 * k] << 8) | data[2 * k + 1];
      }
    } else if (memcmp(type, "IDAT", 4) == 0) {
      idat.insert(idat.end(), data, data + len);
    } else if (memcmp(type, "IEND", 4) == 0) {
      break;
    }
    at += 12 + (size_t)len;
  }
  if (!have_hdr || width == 0 || height == 0 || width > 65536u || height > 65536u) { err = "PNG header missing or size out of range"; return false; }
  if (interlace != 0) { err = "interlaced PNG"; return false; }
  uint32_t channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: err = "PNG colour type"; return false;
  }
  if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))) || (ctype == 3 && depth == 16)) {
    err = "PNG bit depth";
    return false;
  }
  const size_t bpp_bits = (size_t)channels * depth;
  const size_t stride = ((size_t)width * bpp_bits + 7) / 8;
  const size_t bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;  // filter distance in bytes
  if (idat.size() < 6) { err = "PNG without image data"; return false; }
  Bytes raw;
  // zlib wrapper: two header bytes, deflate stream, four check bytes
  if (!inflate(idat.data() + 2, idat.size() - 2, raw, (stride + 1) * (size_t)height) || raw.size() != (stride + 1) * (size_t)height) {
    err = "PNG image data does not inflate to the image";
    return false;
  }
  Bytes prev(stride, 0), cur(stride);
  texels.resize((size_t)width * height);
  for (uint32_t y = 0; y < height; ++y) {
    const unsigned char *line = &raw[(stride + 1) * (size_t)y];
    const unsigned filter = line[0];
    for (size_t i = 0; i < stride; ++i) {
      const int x = line[1 + i];
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int v;
      switch (filter) {
        case 0: v = x; break;
        case 1: v = x + a; break;
        case 2: v = x + b; break;
        case 3: v = x + ((a + b) >> 1); break;
        case 4: {
          const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
          v = x + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
          break;
        }
        default: err = "PNG filter type"; return false;
      }
      cur[i] = (unsigned char)v;
    }
    for (uint32_t x = 0; x < width; ++x) {
      uint32_t s[4] = {0, 0, 0, 0};  // samples at 16-bit scale for the colour-key test, 8-bit values in v8
      unsigned char v8[4] = {0, 0, 0, 255};
      if (depth >= 8) {
        const size_t o = (size_t)x * channels * (depth / 8);
        for (uint32_t k = 0; k < channels; ++k) {
          s[k] = depth == 16 ? (((uint32_t)cur[o + 2 * k] << 8) | cur[o + 2 * k + 1]) : cur[o + k];
          v8[k] = depth == 16 ? cur[o + 2 * k] : cur[o + k];
        }
      } else {
        const size_t bit = (size_t)x * depth;
        const uint32_t v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
        s[0] = v;
        v8[0] = ctype == 3 ? (unsigned char)v : (unsigned char)(v * 255u / ((1u << depth) - 1u));
      }
      unsigned char r, g, bl, al = 255;
      if (ctype == 3) {
        const unsigned char *pe = palette[v8[0]];
        r = pe[0]; g = pe[1]; bl = pe[2]; al = pe[3];
        if (v8[0] >= n_palette) { r = g = bl = 0; }
      } else if (ctype == 0) {
        r = g = bl = v8[0];
        if (have_trns_key && s[0] == trns_key[0]) al = 0;
      } else if (ctype == 4) {
        r = g = bl = v8[0];
        al = v8[1];
      } else if (ctype == 2) {
        r = v8[0]; g = v8[1]; bl = v8[2];
        if (have_trns_key && s[0] == trns_key[0] && s[1] == trns_key[1] && s[2] == trns_key[2]) al = 0;
      } else {
        r = v8[0]; g = v8[1]; bl = v8[2]; al = v8[3];
      }
      texels[(size_t)y * width + x] = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)bl << 16) | ((uint32_t)al << 24);
    }
    prev.swap(cur);
  }
  return true;
}

// ------------------------------------------------------------------ the document
struct Doc {
  Json root;
  std::string dir;             // with a trailing slash, or empty
  std::vector<Bytes> buffers;  // loaded on demand
  std::vector<char> buffer_state;  // 0 = not tried, 1 = loaded, 2 = failed
  Bytes glb_bin;
  bool have_glb_bin = false;
  std::string err;

  const Json *at(const char *array, long i) const {
    const Json *a = root.get(array);
    if (!a || i < 0 || (size_t)i >= a->size()) return nullptr;
    return &a->arr[(size_t)i];
  }
  const Bytes *buffer(long i) {
    const Json *b = at("buffers", i);
    if (!b) { err = "buffer index out of range"; return nullptr; }
    if (buffer_state.size() < root.get("buffers")->size()) {
      buffer_state.resize(root.get("buffers")->size(), 0);
      buffers.resize(buffer_state.size());
    }
    if (buffer_state[(size_t)i] == 1) return &buffers[(size_t)i];
    if (buffer_state[(size_t)i] == 2) return nullptr;
    const Json *uri = b->get("uri");
    bool ok;
    if (!uri) {
      ok = i == 0 && have_glb_bin;
      if (ok) buffers[0] = glb_bin;
      else err = "buffer without uri outside a GLB";
    } else {
      ok = uri->kind == Json::Str && load_uri(uri->str, dir, buffers[(size_t)i], err);
    }
    buffer_state[(size_t)i] = ok ? 1 : 2;
    return ok ? &buffers[(size_t)i] : nullptr;
  }
  // the bytes of a buffer view: pointer, length, stride (0 = tightly packed)
  bool view(long i, const unsigned char *&p, size_t &len, size_t &stride) {
    const Json *v = at("bufferViews", i);
    if (!v) { err = "bufferView index out of range"; return false; }
    const Bytes *b = buffer(v->index("buffer"));
    if (!b) return false;
    const double off = v->number("byteOffset", 0.0), n = v->number("byteLength", -1.0);
    if (off < 0 || n < 0 || off + n > (double)b->size()) { err = "bufferView runs past its buffer"; return false; }
    p = b->data() + (size_t)off;
    len = (size_t)n;
    stride = (size_t)v->number("byteStride", 0.0);
    return true;
  }
};

int components_of(const std::string &type) {
  if (type == "SCALAR") return 1;
  if (type == "VEC2") return 2;
  if (type == "VEC3") return 3;
  if (type == "VEC4") return 4;
  if (type == "MAT4") return 16;
  return 0;
}
size_t component_bytes(long ct) {
  switch (ct) {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
    default: return 0;
  }
}

// An accessor as floats (normalised integers divided by their range when the accessor says so)
// or as unsigned integers.  `want_comp` components per element.
bool read_accessor(Doc &d, long index, int want_comp, bool as_index, std::vector<float> *f, std::vector<uint32_t> *u, size_t &count) {
  const Json *a = d.at("accessors", index);
  if (!a) { d.err = "accessor index out of range"; return false; }
  if (a->has("sparse")) { d.err = "sparse accessor"; return false; }
  const Json *type = a->get("type");
  const int comp = type && type->kind == Json::Str ? components_of(type->str) : 0;
  const long ct = a->index("componentType");
  const size_t cb = component_bytes(ct);
  const double cnt = a->number("count", -1.0);
  if (comp != want_comp || cb == 0 || cnt < 0 || cnt > 4.0e9) { d.err = "accessor has the wrong shape"; return false; }
  count = (size_t)cnt;
  if (as_index && !(ct == 5121 || ct == 5123 || ct == 5125)) { d.err = "index accessor must be unsigned"; return false; }
  const long bv = a->index("bufferView");
  const unsigned char *p = nullptr;
  size_t len = 0, stride = 0;
  const size_t elem = cb * (size_t)comp;
  const double off = a->number("byteOffset", 0.0);
  if (bv >= 0) {
    // the accessor must lie inside its view before anything is sized by its count
    if (!d.view(bv, p, len, stride)) return false;
    if (stride == 0) stride = elem;
    if (off < 0 || (count > 0 && off + (double)(count - 1) * (double)stride + (double)elem > (double)len)) {
      d.err = "accessor runs past its bufferView";
      return false;
    }
  } else if (count > ((size_t)1 << 26)) {
    d.err = "accessor without a bufferView is too large";
    return false;
  }
  if (f) f->assign(count * (size_t)comp, 0.0f);
  if (u) u->assign(count * (size_t)comp, 0u);
  if (bv < 0) return true;  // all zeros
  const Json *nrm = a->get("normalized");
  const bool normalized = nrm && nrm->kind == Json::Bool && nrm->b;
  for (size_t i = 0; i < count; ++i) {
    const unsigned char *e = p + (size_t)off + i * stride;
    for (int k = 0; k < comp; ++k) {
      const unsigned char *c = e + (size_t)k * cb;
      double v;
      switch (ct) {
        case 5120: v = (signed char)c[0]; if (normalized) v = v / 127.0 < -1.0 ? -1.0 : v / 127.0; break;
        case 5121: v = c[0]; if (normalized) v /= 255.0; break;
        case 5122: { int16_t s; memcpy(&s, c, 2); v = s; if (normalized) v = v / 32767.0 < -1.0 ? -1.0 : v / 32767.0; break; }
        case 5123: { uint16_t s; memcpy(&s, c, 2); v = s; if (normalized) v /= 65535.0; break; }
        case 5125: { uint32_t s; memcpy(&s, c, 4); v = s; break; }
        default: { float s; memcpy(&s, c, 4); v = s; break; }
      }
      if (f) (*f)[i * (size_t)comp + (size_t)k] = (float)v;
      if (u) (*u)[i * (size_t)comp + (size_t)k] = (uint32_t)v;
    }
  }
  return true;
}

// ------------------------------------------------------------------ matrices (column-major, as glTF writes them)
struct Mat4 {
  double m[16];
  static Mat4 identity() {
    Mat4 r;
    for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return r;
  }
  double at(int row, int col) const { return m[col * 4 + row]; }
};
Mat4 mul(const Mat4 &a, const Mat4 &b) {
  Mat4 r;
  for (int c = 0; c < 4; ++c)
    for (int rr = 0; rr < 4; ++rr) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += a.at(rr, k) * b.at(k, c);
      r.m[c * 4 + rr] = s;
    }
  return r;
}
Mat4 node_matrix(const Json &node) {
  const Json *mj = node.get("matrix");
  if (mj && mj->size() == 16) {
    Mat4 r;
    for (int i = 0; i < 16; ++i) r.m[i] = mj->arr[(size_t)i].kind == Json::Num ? mj->arr[(size_t)i].num : 0.0;
    return r;
  }
  double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
  auto read = [&](const char *key, double *out, size_t n) {
    const Json *j = node.get(key);
    if (j && j->size() == n)
      for (size_t i = 0; i < n; ++i)
        if (j->arr[i].kind == Json::Num) out[i] = j->arr[i].num;
  };
  read("translation", t, 3);
  read("rotation", q, 4);
  read("scale", s, 3);
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                          {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                          {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
  Mat4 r = Mat4::identity();
  for (int row = 0; row < 3; ++row)
    for (int col = 0; col < 3; ++col) r.m[col * 4 + row] = R[row][col] * s[col];
  r.m[12] = t[0];
  r.m[13] = t[1];
  r.m[14] = t[2];
  return r;
}

// W = T · (Rx · Ry · Rz) · S without shear?  Then the Transform that builds it (src/transform.rs as
// the instance matrices use it: R = Rx · (Ry · Rz), angles about x, y, z).
bool decompose(const Mat4 &w, MirtTransform &out) {
  if (std::fabs(w.at(3, 0)) > 1e-9 || std::fabs(w.at(3, 1)) > 1e-9 || std::fabs(w.at(3, 2)) > 1e-9 || std::fabs(w.at(3, 3) - 1.0) > 1e-9)
    return false;
  double s[3], R[3][3];
  for (int c = 0; c < 3; ++c) {
    s[c] = std::sqrt(w.at(0, c) * w.at(0, c) + w.at(1, c) * w.at(1, c) + w.at(2, c) * w.at(2, c));
    if (!(s[c] > 1e-12) || !std::isfinite(s[c])) return false;
  }
  const double det = w.at(0, 0) * (w.at(1, 1) * w.at(2, 2) - w.at(1, 2) * w.at(2, 1)) -
                     w.at(0, 1) * (w.at(1, 0) * w.at(2, 2) - w.at(1, 2) * w.at(2, 0)) +
                     w.at(0, 2) * (w.at(1, 0) * w.at(2, 1) - w.at(1, 1) * w.at(2, 0));
  if (det < 0.0) s[0] = -s[0];  // a mirror: one negative scale
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R[r][c] = w.at(r, c) / s[c];
  for (int a = 0; a < 3; ++a)
    for (int b = a; b < 3; ++b) {
      const double dot = R[0][a] * R[0][b] + R[1][a] * R[1][b] + R[2][a] * R[2][b];
      if (std::fabs(dot - (a == b ? 1.0 : 0.0)) > 1e-5) return false;  // sheared
    }
  // R = Rx(a) Ry(b) Rz(c):  R02 = sin b,  R12 = -sin a cos b,  R22 = cos a cos b,  R01 = -cos b sin c,  R00 = cos b cos c
  double ax, ay, az;
  const double sb = R[0][2] > 1.0 ? 1.0 : (R[0][2] < -1.0 ? -1.0 : R[0][2]);
  ay = std::asin(sb);
  if (std::fabs(sb) < 1.0 - 1e-9) {
    ax = std::atan2(-R[1][2], R[2][2]);
    az = std::atan2(-R[0][1], R[0][0]);
  } else {  // gimbal lock: all of the remaining rotation goes to x
    az = 0.0;
    ax = std::atan2(R[2][1], R[1][1]);
  }
  memset(&out, 0, sizeof(out));
  out.scale[0] = (float)s[0]; out.scale[1] = (float)s[1]; out.scale[2] = (float)s[2]; out.scale[3] = 1.0f;
  out.rotation[0] = (float)ax; out.rotation[1] = (float)ay; out.rotation[2] = (float)az;
  out.translation[0] = (float)w.m[12]; out.translation[1] = (float)w.m[13]; out.translation[2] = (float)w.m[14];
  return true;
}

MirtTransform identity_transform() {
  MirtTransform t;
  memset(&t, 0, sizeof(t));
  t.scale[0] = t.scale[1] = t.scale[2] = t.scale[3] = 1.0f;
  return t;
}

// ------------------------------------------------------------------ the loader
// An sRGB-encoded 8-bit colour channel as the nearest linear 8-bit value (IEC 61966-2-1's piecewise curve).
uint32_t srgb_to_linear_rgba8(uint32_t px) {
  static const std::vector<unsigned char> table = [] {
    std::vector<unsigned char> t(256);
    for (int i = 0; i < 256; ++i) {
      const double c = i / 255.0;
      const double lin = c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
      t[(size_t)i] = (unsigned char)std::lround(lin * 255.0);
    }
    return t;
  }();
  return (px & 0xFF000000u) | ((uint32_t)table[(px >> 16) & 0xFF] << 16) | ((uint32_t)table[(px >> 8) & 0xFF] << 8) |
         (uint32_t)table[px & 0xFF];
}

struct Loader {
  Scene &scene;
  Doc &d;
  MirthGltfInfo info;
  std::string key;                       // descriptor prefix for the MeshServer
  std::map<long, int> texture_ids;       // glTF texture -> 1-based texture id, 0 = unusable
  std::map<long, uint32_t> material_ids; // glTF material (-1 = none) -> MaterialServer id
  struct Prim {
    bool ok = false;
    Mesh mesh;
    std::vector<float> uvs;
    long material = -1;
    bool file_normals = false;         // NORMAL came from the file (else face-derived)
    uint32_t shared_id = 0xFFFFFFFFu;  // MeshServer id of the unbaked mesh, once some node used it
  };
  std::vector<char> visited;  // per node: already instanced (glTF's node graph is a strict tree)
  std::map<std::pair<long, long>, Prim> prims;
  uint32_t n_baked = 0;

  Loader(Scene &s, Doc &doc, const std::string &k) : scene(s), d(doc), key(k) { memset(&info, 0, sizeof(info)); }

  int texture(const Json *ref) {
    if (!ref) return 0;
    const long ti = ref->index("index");
    if (ti < 0) return 0;
    auto it = texture_ids.find(ti);
    if (it != texture_ids.end()) return it->second;
    int id = 0;
    const Json *t = d.at("textures", ti);
    const Json *img = t ? d.at("images", t->index("source")) : nullptr;
    if (img) {
      Bytes bytes;
      std::string err;
      bool have = false;
      const Json *uri = img->get("uri");
      if (uri && uri->kind == Json::Str) {
        have = load_uri(uri->str, d.dir, bytes, err);
      } else if (img->index("bufferView") >= 0) {
        const unsigned char *p;
        size_t len, stride;
        if (d.view(img->index("bufferView"), p, len, stride)) {
          bytes.assign(p, p + len);
          have = true;
        }
      }
      uint32_t w = 0, h = 0;
      std::vector<uint32_t> texels;
      if (have && decode_png(bytes, w, h, texels, err) && scene.texels.size() + texels.size() <= 0xFFFFFFFFull) {
        // glTF defines baseColorTexture and emissiveTexture (the only two slots read here) as sRGB-encoded; the
        // sampler takes texels as linear channel / 255, so colour channels are decoded here, alpha stays
        for (uint32_t &px : texels) px = srgb_to_linear_rgba8(px);
        uint32_t filter = MIRT_FILTER_BILINEAR;
        const Json *smp = d.at("samplers", t->index("sampler"));
        if (smp && smp->index("magFilter") == 9728) filter = MIRT_FILTER_NEAREST;
        MirtTexture mt{w, h, (uint32_t)scene.texels.size(), filter};
        scene.texels.insert(scene.texels.end(), texels.begin(), texels.end());
        scene.textures.push_back(mt);
        id = (int)scene.textures.size();
        info.n_textures++;
      } else {
        info.n_skipped_images++;
      }
    } else {
      info.n_skipped_images++;
    }
    texture_ids[ti] = id;
    return id;
  }

  uint32_t material(long mi) {
    auto it = material_ids.find(mi);
    if (it != material_ids.end()) return it->second;
    MirtMaterial m = default_material();
    // glTF's defaults: white, fully metallic, fully rough
    m.colour[0] = m.colour[1] = m.colour[2] = m.colour[3] = 1.0f;
    m.metallic = 1.0f;
    m.roughness = 1.0f;
    const Json *j = d.at("materials", mi);
    if (j) {
      if (const Json *pbr = j->get("pbrMetallicRoughness")) {
        const Json *c = pbr->get("baseColorFactor");
        if (c && c->size() == 4)
          for (size_t k = 0; k < 4; ++k)
            if (c->arr[k].kind == Json::Num) m.colour[k] = (float)c->arr[k].num;
        m.metallic = (float)pbr->number("metallicFactor", 1.0);
        m.roughness = (float)pbr->number("roughnessFactor", 1.0);
        m.colour_texture = (uint32_t)texture(pbr->get("baseColorTexture"));
      }
      double strength = 1.0;
      if (const Json *ext = j->get("extensions")) {
        if (const Json *e = ext->get("KHR_materials_emissive_strength")) strength = e->number("emissiveStrength", 1.0);
        if (const Json *e = ext->get("KHR_materials_ior")) m.ior = (float)e->number("ior", 1.5);
        if (const Json *e = ext->get("KHR_materials_transmission")) m.transmission = (float)e->number("transmissionFactor", 0.0);
      }
      const Json *e = j->get("emissiveFactor");
      if (e && e->size() == 3) {
        for (size_t k = 0; k < 3; ++k)
          if (e->arr[k].kind == Json::Num) m.emissive[k] = (float)(e->arr[k].num * strength);
        m.emissive[3] = 1.0f;
      }
      m.emissive_texture = (uint32_t)texture(j->get("emissiveTexture"));
    }
    const uint32_t id = scene.add_material(m);
    material_ids[mi] = id;
    info.n_materials++;
    return id;
  }

  Prim &primitive(long mesh_i, long prim_i, const Json &p) {
    Prim &out = prims[{mesh_i, prim_i}];
    if (out.ok || out.material != -1) return out;
    out.material = -2;  // tried
    const long mode = p.has("mode") ? p.index("mode") : 4;
    const Json *attrs = p.get("attributes");
    if (mode != 4 || !attrs || attrs->index("POSITION") < 0 || p.has("targets")) {
      info.n_skipped_primitives++;
      return out;
    }
    std::vector<float> pos, nrm, uv;
    std::vector<uint32_t> idx;
    size_t n_pos = 0, n = 0;
    if (!read_accessor(d, attrs->index("POSITION"), 3, false, &pos, nullptr, n_pos) || n_pos == 0) {
      info.n_skipped_primitives++;
      return out;
    }
    if (p.index("indices") >= 0) {
      if (!read_accessor(d, p.index("indices"), 1, true, nullptr, &idx, n)) {
        info.n_skipped_primitives++;
        return out;
      }
    } else {
      idx.resize(n_pos);
      for (size_t i = 0; i < n_pos; ++i) idx[i] = (uint32_t)i;
    }
    idx.resize(idx.size() / 3 * 3);
    for (uint32_t i : idx)
      if (i >= n_pos) {
        d.err = "index out of range in a primitive";
        info.n_skipped_primitives++;
        return out;
      }
    if (idx.empty()) {
      info.n_skipped_primitives++;
      return out;
    }
    const bool have_n = attrs->index("NORMAL") >= 0 && read_accessor(d, attrs->index("NORMAL"), 3, false, &nrm, nullptr, n) && n == n_pos;
    const bool have_uv = attrs->index("TEXCOORD_0") >= 0 && read_accessor(d, attrs->index("TEXCOORD_0"), 2, false, &uv, nullptr, n) && n == n_pos;
    Mesh &m = out.mesh;
    m.positions.resize(n_pos);
    for (size_t i = 0; i < n_pos; ++i) m.positions[i] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], 1.0f};
    m.faces.reserve(idx.size() / 3 * 4);
    for (size_t t = 0; t + 2 < idx.size(); t += 3) {
      m.faces.push_back(idx[t]);
      m.faces.push_back(idx[t + 1]);
      m.faces.push_back(idx[t + 2]);
      m.faces.push_back(0);
    }
    if (have_n) {
      m.normals.resize(n_pos);
      for (size_t i = 0; i < n_pos; ++i) m.normals[i] = {nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2], 0.0f};
    } else {
      m.normals = compute_vertex_normals_ccw(m.positions, idx.data(), idx.size());
    }
    if (have_uv) out.uvs = uv;
    out.file_normals = have_n;
    out.material = p.has("material") ? p.index("material") : -1;
    if (out.material < -1) out.material = -1;
    out.ok = true;
    return out;
  }

  uint32_t add(const std::string &desc, const Mesh &mesh, const std::vector<float> &uvs) {
    const uint32_t id = scene.add_mesh(desc, mesh);
    if (id != 0xFFFFFFFFu && !uvs.empty() && scene.meshes[id].uvs.empty()) scene.meshes[id].uvs = uvs;
    return id;
  }

  bool node(long ni, const Mat4 &parent, int depth) {
    const Json *n = d.at("nodes", ni);
    if (!n || depth > 256) return true;  // a bad index, or a chain deeper than any real file's: ignored
    // glTF requires a strict tree (a node has at most one parent and is no ancestor of itself).  A node reached a
    // second time — a cycle, or a diamond, which with a fan-out of two doubles the visits per level — is skipped,
    // so a file is walked in time linear in its node count.
    if (visited.empty()) {
      const Json *all = d.root.get("nodes");
      visited.assign(all ? all->size() : 0, 0);
    }
    if ((size_t)ni >= visited.size() || visited[(size_t)ni]) return true;
    visited[(size_t)ni] = 1;
    const Mat4 world = mul(parent, node_matrix(*n));
    const long mi = n->index("mesh");
    const Json *mesh = d.at("meshes", mi);
    const Json *plist = mesh ? mesh->get("primitives") : nullptr;
    for (size_t pi = 0; plist && pi < plist->size(); ++pi) {
      Prim &p = primitive(mi, (long)pi, plist->arr[pi]);
      if (!p.ok) continue;
      const uint32_t mat = material(p.material);
      MirtTransform t;
      uint32_t mesh_id;
      if (decompose(world, t)) {
        if (p.shared_id == 0xFFFFFFFFu) {
          p.shared_id = add(key + "#m" + std::to_string(mi) + "p" + std::to_string(pi), p.mesh, p.uvs);
          if (p.shared_id == 0xFFFFFFFFu) return false;
          info.n_meshes++;
        }
        mesh_id = p.shared_id;
      } else {
        // a sheared world matrix: a copy of the mesh with the matrix applied to the positions; normals the file
        // gave go through the inverse transpose (the cofactor matrix, signed by the determinant) and are
        // renormalised, face-derived ones are derived again from the moved triangles
        Mesh baked = p.mesh;
        for (Vec4 &v : baked.positions) {
          const double x = v.x, y = v.y, z = v.z;
          v = {(float)(world.at(0, 0) * x + world.at(0, 1) * y + world.at(0, 2) * z + world.at(0, 3)),
               (float)(world.at(1, 0) * x + world.at(1, 1) * y + world.at(1, 2) * z + world.at(1, 3)),
               (float)(world.at(2, 0) * x + world.at(2, 1) * y + world.at(2, 2) * z + world.at(2, 3)), 1.0f};
        }
        std::vector<uint32_t> idx;
        for (size_t f = 0; f < baked.n_faces(); ++f) {
          idx.push_back(baked.faces[4 * f]);
          idx.push_back(baked.faces[4 * f + 1]);
          idx.push_back(baked.faces[4 * f + 2]);
        }
        if (p.file_normals) {
          double c[3][3];
          for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) {
              const int r1 = (r + 1) % 3, r2 = (r + 2) % 3, q1 = (q + 1) % 3, q2 = (q + 2) % 3;
              c[r][q] = world.at(r1, q1) * world.at(r2, q2) - world.at(r1, q2) * world.at(r2, q1);
            }
          const double det = world.at(0, 0) * c[0][0] + world.at(0, 1) * c[0][1] + world.at(0, 2) * c[0][2];
          const double sign = det < 0.0 ? -1.0 : 1.0;
          for (Vec4 &nv : baked.normals) {
            const double x = nv.x, y = nv.y, z = nv.z;
            double o[3];
            for (int r = 0; r < 3; ++r) o[r] = sign * (c[r][0] * x + c[r][1] * y + c[r][2] * z);
            const double len = std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
            if (len > 0.0) nv = {(float)(o[0] / len), (float)(o[1] / len), (float)(o[2] / len), 0.0f};
          }
        } else {
          baked.normals = compute_vertex_normals_ccw(baked.positions, idx.data(), idx.size());
        }
        mesh_id = add(key + "#m" + std::to_string(mi) + "p" + std::to_string(pi) + "@n" + std::to_string(ni) + "." + std::to_string(n_baked++),
                      baked, p.uvs);
        if (mesh_id == 0xFFFFFFFFu) return false;
        info.n_meshes++;
        info.n_baked++;
        t = identity_transform();
      }
      scene.spawn(t, mesh_id, mat);
      info.n_instances++;
    }
    const Json *kids = n->get("children");
    for (size_t k = 0; kids && k < kids->size(); ++k)
      if (kids->arr[k].as_index() >= 0 && !node(kids->arr[k].as_index(), world, depth + 1)) return false;
    return true;
  }
};

}  // namespace

bool load_gltf(Scene &scene, const std::string &path, MirthGltfInfo &info, std::string &err) {
  Bytes file;
  if (!read_file(path, file, err)) return false;
  Doc d;
  const size_t slash = path.find_last_of('/');
  d.dir = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
  const char *json = reinterpret_cast<const char *>(file.data());
  size_t json_len = file.size();
  if (file.size() >= 12 && memcmp(file.data(), "glTF", 4) == 0) {
    // GLB: header (magic, version, length), then chunks (length, type, data): JSON first, BIN optional
    auto le32 = [&](size_t at) { return (uint32_t)file[at] | ((uint32_t)file[at + 1] << 8) | ((uint32_t)file[at + 2] << 16) | ((uint32_t)file[at + 3] << 24); };
    if (le32(4) != 2u) { err = "GLB version must be 2"; return false; }
    size_t at = 12;
    bool have_json = false;
    while (at + 8 <= file.size()) {
      const size_t len = le32(at);
      const uint32_t type = le32(at + 4);
      if (len > file.size() - at - 8) { err = "GLB chunk runs past the file"; return false; }
      if (type == 0x4E4F534Au && !have_json) {
        json = reinterpret_cast<const char *>(file.data()) + at + 8;
        json_len = len;
        have_json = true;
      } else if (type == 0x004E4942u && !d.have_glb_bin) {
        d.glb_bin.assign(file.begin() + (long)(at + 8), file.begin() + (long)(at + 8 + len));
        d.have_glb_bin = true;
      }
      at += 8 + ((len + 3) & ~(size_t)3);
    }
    if (!have_json) { err = "GLB without a JSON chunk"; return false; }
  }
  JsonParser jp{json, json + json_len, std::string(), 0};
  if (!jp.value(d.root) || d.root.kind != Json::Obj) {
    err = "glTF: " + (jp.err.empty() ? std::string("not a JSON object") : jp.err);
    return false;
  }
  const Json *asset = d.root.get("asset");
  const Json *version = asset ? asset->get("version") : nullptr;
  if (!version || version->kind != Json::Str || version->str.compare(0, 2, "2.") != 0) {
    err = "glTF: asset.version must be 2.x";
    return false;
  }
  Loader L(scene, d, "GLTF:" + path);
  // the default scene's roots; a file without scenes: every node that is nobody's child
  std::vector<long> roots;
  const Json *scenes = d.root.get("scenes");
  if (scenes && scenes->size() > 0) {
    long si = d.root.index("scene");
    if (si < 0 || (size_t)si >= scenes->size()) si = 0;
    const Json *nodes = scenes->arr[(size_t)si].get("nodes");
    for (size_t i = 0; nodes && i < nodes->size(); ++i)
      if (nodes->arr[i].as_index() >= 0) roots.push_back(nodes->arr[i].as_index());
  } else if (const Json *nodes = d.root.get("nodes")) {
    std::vector<char> is_child(nodes->size(), 0);
    for (const Json &n : nodes->arr) {
      const Json *kids = n.get("children");
      for (size_t k = 0; kids && k < kids->size(); ++k)
        if (kids->arr[k].as_index() >= 0 && (size_t)kids->arr[k].as_index() < is_child.size())
          is_child[(size_t)kids->arr[k].as_index()] = 1;
    }
    for (size_t i = 0; i < is_child.size(); ++i)
      if (!is_child[i]) roots.push_back((long)i);
  }
  for (long r : roots)
    if (!L.node(r, Mat4::identity(), 0)) {
      err = scene.build_error.empty() ? "glTF: a mesh could not be added" : scene.build_error;
      return false;
    }
  if (L.info.n_instances == 0) {
    err = "glTF: no triangle primitive in the scene" + (d.err.empty() ? std::string() : " (" + d.err + ")");
    return false;
  }
  info = L.info;
  return true;
}

bool decode_png_rgba8(const unsigned char *png, size_t len, uint32_t &width, uint32_t &height, std::vector<uint32_t> &texels,
                      std::string &err) {
  Bytes b(png, png + len);
  return decode_png(b, width, height, texels, err);
}

}  // namespace mirth
