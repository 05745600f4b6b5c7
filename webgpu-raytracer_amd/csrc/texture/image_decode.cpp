// image_decode.cpp — host-side decode of the encoded images a scene carries (include/mi355tex.h).
//
// The reference leaves this step to the browser: createImageBitmap(new Blob([data]), ...) in
// /root/reference/src/renderer/ResourceManager.ts:162-176.  Nothing of a browser exists here, so the two containers
// glTF allows (PNG, JPEG) are decoded from their specifications:
//   inflate   RFC 1951 (stored / fixed / dynamic blocks) inside the RFC 1950 zlib wrapper (Adler-32 checked)
//   PNG       ISO/IEC 15948: all colour types and bit depths, Adam7, tRNS, the five filters, CRC-32 checked
//   JPEG      ITU-T T.81 Huffman, sequential and progressive (SOF0 / SOF1 / SOF2, 8 bit), JFIF / Adobe colour, restart
//             intervals;
//             integer IDCT, triangle chroma upsampling and fixed-point YCbCr -> RGB follow the algorithms the IJG
//             library documents (jidctint / jdsample "fancy" / jdcolor), so output equals the common decoders' bit for bit
// No GPU, no library dependency; plain C ABI.  Written from the specifications; the canonical-Huffman construction and
// decode loop of the inflater have the shape of zlib's contrib puff.c (Mark Adler), the usual way to write them.
#include "../../../include/mi355tex.h"

#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace {

thread_local std::string g_error;

int fail(int code, const char* msg) {
  g_error = msg;
  return code;
}

// ------------------------------------------------------------------------------------------- inflate (RFC 1951)
struct BitReader {
  const uint8_t* p;
  size_t n, pos;
  uint32_t bitbuf;
  int bitcnt;
  bool overrun;
  int bits(int need) {  // need <= 16, LSB first
    while (bitcnt < need) {
      if (pos >= n) {
        overrun = true;
        return 0;
      }
      bitbuf |= (uint32_t)p[pos++] << bitcnt;
      bitcnt += 8;
    }
    int v = (int)(bitbuf & ((1u << need) - 1u));
    bitbuf >>= need;
    bitcnt -= need;
    return v;
  }
};

struct Huffman {
  uint16_t count[16];
  uint16_t symbol[288];
};

// canonical code from code lengths; returns 0 complete, >0 incomplete, <0 over-subscribed
int build_huffman(Huffman& h, const uint8_t* length, int n) {
  for (int i = 0; i < 16; i++) h.count[i] = 0;
  for (int i = 0; i < n; i++) h.count[length[i]]++;
  if (h.count[0] == n) return 0;
  int left = 1;
  for (int len = 1; len < 16; len++) {
    left <<= 1;
    left -= h.count[len];
    if (left < 0) return left;
  }
  uint16_t offs[16];
  offs[1] = 0;
  for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + h.count[len];
  for (int i = 0; i < n; i++)
    if (length[i] != 0) h.symbol[offs[length[i]]++] = (uint16_t)i;
  return left;
}

int decode_symbol(BitReader& br, const Huffman& h) {
  int code = 0, first = 0, index = 0;
  for (int len = 1; len < 16; len++) {
    code |= br.bits(1);
    if (br.overrun) return -1;
    int count = h.count[len];
    if (code - count < first) return h.symbol[index + (code - first)];
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint16_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint16_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

int inflate_codes(BitReader& br, std::vector<uint8_t>& out, size_t cap, const Huffman& lencode, const Huffman& distcode) {
  for (;;) {
    int sym = decode_symbol(br, lencode);
    if (sym < 0) return -1;
    if (sym < 256) {
      if (out.size() >= cap) return -2;
      out.push_back((uint8_t)sym);
    } else if (sym == 256) {
      return 0;
    } else {
      sym -= 257;
      if (sym >= 29) return -1;
      int len = kLenBase[sym] + br.bits(kLenExtra[sym]);
      int ds = decode_symbol(br, distcode);
      if (ds < 0 || ds >= 30) return -1;
      size_t dist = (size_t)kDistBase[ds] + (size_t)br.bits(kDistExtra[ds]);
      if (br.overrun || dist > out.size()) return -1;
      if (out.size() + (size_t)len > cap) return -2;
      size_t from = out.size() - dist;
      for (int i = 0; i < len; i++) out.push_back(out[from + (size_t)i]);
    }
  }
}

// raw deflate stream -> out (at most cap bytes); 0 ok, -1 corrupt, -2 output limit
int inflate_raw(BitReader& br, std::vector<uint8_t>& out, size_t cap) {
  static Huffman fixed_len, fixed_dist;
  static bool fixed_ready = false;
  if (!fixed_ready) {
    uint8_t l[288];
    for (int i = 0; i < 144; i++) l[i] = 8;
    for (int i = 144; i < 256; i++) l[i] = 9;
    for (int i = 256; i < 280; i++) l[i] = 7;
    for (int i = 280; i < 288; i++) l[i] = 8;
    build_huffman(fixed_len, l, 288);
    for (int i = 0; i < 30; i++) l[i] = 5;
    build_huffman(fixed_dist, l, 30);
    fixed_ready = true;
  }
  int last;
  do {
    last = br.bits(1);
    int type = br.bits(2);
    if (br.overrun) return -1;
    if (type == 0) {
      br.bitbuf = 0;
      br.bitcnt = 0;
      if (br.pos + 4 > br.n) return -1;
      unsigned len = br.p[br.pos] | (br.p[br.pos + 1] << 8);
      unsigned nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
      br.pos += 4;
      if ((len ^ 0xffffu) != nlen || br.pos + len > br.n) return -1;
      if (out.size() + len > cap) return -2;
      out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
      br.pos += len;
    } else if (type == 1) {
      int r = inflate_codes(br, out, cap, fixed_len, fixed_dist);
      if (r) return r;
    } else if (type == 2) {
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      int nlen = br.bits(5) + 257, ndist = br.bits(5) + 1, ncode = br.bits(4) + 4;
      if (br.overrun || nlen > 286 || ndist > 30) return -1;
      uint8_t lengths[320];
      memset(lengths, 0, sizeof lengths);
      for (int i = 0; i < ncode; i++) lengths[order[i]] = (uint8_t)br.bits(3);
      Huffman cl;
      if (build_huffman(cl, lengths, 19) != 0) return -1;
      uint8_t ll[320];
      int idx = 0;
      while (idx < nlen + ndist) {
        int sym = decode_symbol(br, cl);
        if (sym < 0) return -1;
        if (sym < 16) {
          ll[idx++] = (uint8_t)sym;
        } else {
          int prev = 0, rep;
          if (sym == 16) {
            if (idx == 0) return -1;
            prev = ll[idx - 1];
            rep = 3 + br.bits(2);
          } else if (sym == 17) {
            rep = 3 + br.bits(3);
          } else {
            rep = 11 + br.bits(7);
          }
          if (br.overrun || idx + rep > nlen + ndist) return -1;
          while (rep--) ll[idx++] = (uint8_t)prev;
        }
      }
      if (ll[256] == 0) return -1;
      Huffman lc, dc;
      int e = build_huffman(lc, ll, nlen);
      if (e < 0 || (e > 0 && nlen - lc.count[0] != 1)) return -1;
      e = build_huffman(dc, ll + nlen, ndist);
      if (e < 0 || (e > 0 && ndist - dc.count[0] != 1)) return -1;
      int r = inflate_codes(br, out, cap, lc, dc);
      if (r) return r;
    } else {
      return -1;
    }
  } while (!last);
  return 0;
}

// zlib wrapper (RFC 1950)
int inflate_zlib(const uint8_t* data, size_t size, std::vector<uint8_t>& out, size_t cap) {
  if (size < 6) return -1;
  const unsigned cmf = data[0], flg = data[1];
  if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u || (flg & 0x20u)) return -1;
  BitReader br = {data + 2, size - 2, 0, 0, 0, false};
  int r = inflate_raw(br, out, cap);
  if (r) return r;
  size_t tail = br.pos;  // whole bytes consumed (bits left in bitbuf belong to the last byte read)
  if (tail + 4 > size - 2) return -1;
  const uint8_t* a = data + 2 + tail;
  uint32_t want = ((uint32_t)a[0] << 24) | ((uint32_t)a[1] << 16) | ((uint32_t)a[2] << 8) | a[3];
  uint32_t s1 = 1, s2 = 0;
  for (size_t i = 0; i < out.size(); i++) {
    s1 = (s1 + out[i]) % 65521u;
    s2 = (s2 + s1) % 65521u;
  }
  return ((s2 << 16) | s1) == want ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------------ PNG
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 255u] ^ (crc >> 8);
  return crc;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
  int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

int decode_png(const uint8_t* data, size_t size, mt_image* out) {
  size_t pos = 8;
  uint32_t width = 0, height = 0;
  int depth = 0, ctype = 0, interlace = 0;
  bool have_ihdr = false, have_plte = false, have_trns = false, seen_iend = false;
  uint8_t palette[256][4];
  for (int i = 0; i < 256; i++) palette[i][0] = palette[i][1] = palette[i][2] = 0, palette[i][3] = 255;
  int n_palette = 0;
  uint16_t key[3] = {0, 0, 0};
  std::vector<uint8_t> idat;
  while (pos + 12 <= size && !seen_iend) {
    const uint32_t len = be32(data + pos);
    if (len > 0x7fffffffu || pos + 12 + (size_t)len > size) return fail(MT_ERR_CORRUPT, "png: truncated chunk");
    const uint8_t* type = data + pos + 4;
    const uint8_t* body = data + pos + 8;
    if ((crc32_update(0xffffffffu, type, 4 + (size_t)len) ^ 0xffffffffu) != be32(body + len))
      return fail(MT_ERR_CORRUPT, "png: chunk CRC mismatch");
    if (!have_ihdr && memcmp(type, "IHDR", 4) != 0) return fail(MT_ERR_CORRUPT, "png: IHDR is not first");
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13 || have_ihdr) return fail(MT_ERR_CORRUPT, "png: bad IHDR");
      width = be32(body);
      height = be32(body + 4);
      depth = body[8];
      ctype = body[9];
      interlace = body[12];
      if (width == 0 || height == 0 || width > 32768u || height > 32768u) return fail(MT_ERR_UNSUPPORTED, "png: size out of range");
      if (body[10] != 0 || body[11] != 0 || interlace > 1) return fail(MT_ERR_CORRUPT, "png: bad IHDR methods");
      const bool ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                      ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16)) ||
                      (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8));
      if (!ok) return fail(MT_ERR_CORRUPT, "png: bad colour type / bit depth");
      have_ihdr = true;
    } else if (!memcmp(type, "PLTE", 4)) {
      if (len % 3u != 0u || len > 768u || len == 0u) return fail(MT_ERR_CORRUPT, "png: bad PLTE");
      n_palette = (int)(len / 3u);
      for (int i = 0; i < n_palette; i++) memcpy(palette[i], body + 3 * i, 3);
      have_plte = true;
    } else if (!memcmp(type, "tRNS", 4)) {
      if (ctype == 3) {
        if (!have_plte || len > (uint32_t)n_palette) return fail(MT_ERR_CORRUPT, "png: bad tRNS");
        for (uint32_t i = 0; i < len; i++) palette[i][3] = body[i];
      } else if (ctype == 0) {
        if (len != 2) return fail(MT_ERR_CORRUPT, "png: bad tRNS");
        key[0] = (uint16_t)((body[0] << 8) | body[1]);
      } else if (ctype == 2) {
        if (len != 6) return fail(MT_ERR_CORRUPT, "png: bad tRNS");
        for (int i = 0; i < 3; i++) key[i] = (uint16_t)((body[2 * i] << 8) | body[2 * i + 1]);
      } else {
        return fail(MT_ERR_CORRUPT, "png: tRNS with an alpha colour type");
      }
      have_trns = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!memcmp(type, "IEND", 4)) {
      seen_iend = true;
    } else if (!(type[0] & 0x20u)) {
      return fail(MT_ERR_UNSUPPORTED, "png: unknown critical chunk");
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || !seen_iend || idat.empty()) return fail(MT_ERR_CORRUPT, "png: missing IHDR / IDAT / IEND");
  if (ctype == 3 && !have_plte) return fail(MT_ERR_CORRUPT, "png: palette image without PLTE");

  const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
  const int bits_px = channels * depth;
  const int bpp = bits_px >= 8 ? bits_px / 8 : 1;
  static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
  const int n_pass = interlace ? 7 : 1;
  size_t raw_size = 0;
  for (int ps = 0; ps < n_pass; ps++) {
    const uint32_t pw = interlace ? (width + (uint32_t)dx[ps] - 1u - (uint32_t)xs[ps]) / (uint32_t)dx[ps] : width;
    const uint32_t ph = interlace ? (height + (uint32_t)dy[ps] - 1u - (uint32_t)ys[ps]) / (uint32_t)dy[ps] : height;
    if (pw && ph) raw_size += (size_t)ph * (1 + ((size_t)pw * (size_t)bits_px + 7) / 8);
  }
  std::vector<uint8_t> raw;
  raw.reserve(raw_size);
  int r = inflate_zlib(idat.data(), idat.size(), raw, raw_size);
  if (r == -2) return fail(MT_ERR_CORRUPT, "png: more image data than the header declares");
  if (r || raw.size() != raw_size) return fail(MT_ERR_CORRUPT, "png: corrupt or short image data");

  uint8_t* rgba = (uint8_t*)malloc((size_t)width * height * 4);
  if (!rgba) return fail(MT_ERR_MEMORY, "png: out of memory");
  size_t rp = 0;
  std::vector<uint8_t> prev, cur;
  for (int ps = 0; ps < n_pass; ps++) {
    const uint32_t pw = interlace ? (width + (uint32_t)dx[ps] - 1u - (uint32_t)xs[ps]) / (uint32_t)dx[ps] : width;
    const uint32_t ph = interlace ? (height + (uint32_t)dy[ps] - 1u - (uint32_t)ys[ps]) / (uint32_t)dy[ps] : height;
    if (!pw || !ph) continue;
    const size_t rowbytes = ((size_t)pw * (size_t)bits_px + 7) / 8;
    prev.assign(rowbytes, 0);
    cur.resize(rowbytes);
    for (uint32_t y = 0; y < ph; y++) {
      const int ft = raw[rp++];
      const uint8_t* src = raw.data() + rp;
      rp += rowbytes;
      if (ft > 4) {
        free(rgba);
        return fail(MT_ERR_CORRUPT, "png: bad filter type");
      }
      for (size_t i = 0; i < rowbytes; i++) {
        const int a = i >= (size_t)bpp ? cur[i - (size_t)bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - (size_t)bpp] : 0;
        int v = src[i];
        if (ft == 1) v += a;
        else if (ft == 2) v += b;
        else if (ft == 3) v += (a + b) >> 1;
        else if (ft == 4) v += paeth(a, b, c);
        cur[i] = (uint8_t)v;
      }
      const uint32_t oy = interlace ? (uint32_t)ys[ps] + y * (uint32_t)dy[ps] : y;
      for (uint32_t x = 0; x < pw; x++) {
        const uint32_t ox = interlace ? (uint32_t)xs[ps] + x * (uint32_t)dx[ps] : x;
        uint16_t s[4] = {0, 0, 0, 0};  // raw samples at the file's bit depth
        if (depth == 8) {
          for (int c = 0; c < channels; c++) s[c] = cur[(size_t)x * (size_t)channels + (size_t)c];
        } else if (depth == 16) {
          for (int c = 0; c < channels; c++) {
            const size_t o = ((size_t)x * (size_t)channels + (size_t)c) * 2;
            s[c] = (uint16_t)((cur[o] << 8) | cur[o + 1]);
          }
        } else {
          const size_t bit = (size_t)x * (size_t)depth;
          s[0] = (uint16_t)((cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1 << depth) - 1));
        }
        uint8_t* px = rgba + ((size_t)oy * width + ox) * 4;
        auto to8 = [&](uint16_t v) -> uint8_t {
          if (depth == 16) return (uint8_t)(v >> 8);
          if (depth == 8) return (uint8_t)v;
          return (uint8_t)(v * 255 / ((1 << depth) - 1));
        };
        if (ctype == 3) {
          if (s[0] >= (uint16_t)n_palette) {
            free(rgba);
            return fail(MT_ERR_CORRUPT, "png: palette index out of range");
          }
          memcpy(px, palette[s[0]], 4);
        } else if (ctype == 0) {
          px[0] = px[1] = px[2] = to8(s[0]);
          px[3] = (have_trns && s[0] == key[0]) ? 0 : 255;
        } else if (ctype == 2) {
          px[0] = to8(s[0]);
          px[1] = to8(s[1]);
          px[2] = to8(s[2]);
          px[3] = (have_trns && s[0] == key[0] && s[1] == key[1] && s[2] == key[2]) ? 0 : 255;
        } else if (ctype == 4) {
          px[0] = px[1] = px[2] = to8(s[0]);
          px[3] = to8(s[1]);
        } else {
          px[0] = to8(s[0]);
          px[1] = to8(s[1]);
          px[2] = to8(s[2]);
          px[3] = to8(s[3]);
        }
      }
      prev.swap(cur);
    }
  }
  out->width = width;
  out->height = height;
  out->rgba = rgba;
  return MT_OK;
}

// ----------------------------------------------------------------------------------------------------- JPEG
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct JHuff {
  bool present;
  uint8_t bits[17];
  uint8_t vals[256];
  int mincode[17], maxcode[18], valptr[17];
};

void jhuff_prepare(JHuff& h) {
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    h.valptr[l] = k;
    h.mincode[l] = code;
    code += h.bits[l];
    k += h.bits[l];
    h.maxcode[l] = h.bits[l] ? code - 1 : -1;
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff;
}

struct JBits {
  const uint8_t* p;
  size_t n, pos;
  uint32_t buf;
  int cnt;
  bool hit_marker;
  int bit() {
    if (cnt == 0) {
      int b = 0;
      if (!hit_marker && pos < n) {
        b = p[pos];
        if (b == 0xff) {
          if (pos + 1 < n && p[pos + 1] == 0x00) {
            pos += 2;
          } else {
            hit_marker = true;  // a marker inside entropy data: feed zeros (T.81 F.2.2.5 decoders pad)
            b = 0;
          }
        } else {
          pos++;
        }
      } else {
        hit_marker = true;
      }
      buf = (uint32_t)b;
      cnt = 8;
    }
    cnt--;
    return (int)((buf >> cnt) & 1u);
  }
  int receive(int s) {
    int v = 0;
    for (int i = 0; i < s; i++) v = (v << 1) | bit();
    return v;
  }
};

int jhuff_decode(JBits& br, const JHuff& h) {
  int code = br.bit();
  for (int l = 1; l <= 16; l++) {
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    code = (code << 1) | br.bit();
  }
  return -1;
}

inline int jextend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

struct JComp {
  int id, h, v, tq, td, ta;
  int blocks_w, blocks_h;          // allocated blocks (whole MCUs)
  int width, height;               // component size in samples: ceil(image * h / hmax)
  std::vector<int16_t> coef;       // blocks_w * blocks_h * 64, natural order, not dequantised
  std::vector<uint8_t> plane;      // blocks_w*8 x blocks_h*8 after the IDCT
  int pred;
};

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

#define JFIX_0_298631336 2446
#define JFIX_0_390180644 3196
#define JFIX_0_541196100 4433
#define JFIX_0_765366865 6270
#define JFIX_0_899976223 7373
#define JFIX_1_175875602 9633
#define JFIX_1_501321110 12299
#define JFIX_1_847759065 15137
#define JFIX_1_961570560 16069
#define JFIX_2_053119869 16819
#define JFIX_2_562915447 20995
#define JFIX_3_072711026 25172
inline long jdescale(long x, int n) { return (x + (1L << (n - 1))) >> n; }

// Integer "slow but accurate" inverse DCT (13-bit constants, 2 extra bits between the passes), dequantising on the fly.
void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
  const int CONST_BITS = 13, PASS1_BITS = 2;
  long ws[64];
  for (int c = 0; c < 8; c++) {
    const int16_t* ip = in + c;
    const uint16_t* qp = q + c;
    long* wp = ws + c;
    if (ip[8] == 0 && ip[16] == 0 && ip[24] == 0 && ip[32] == 0 && ip[40] == 0 && ip[48] == 0 && ip[56] == 0) {
      long dc = ((long)ip[0] * qp[0]) * (1L << PASS1_BITS);  // (a shift of a negative value is undefined before C++20)
      for (int r = 0; r < 8; r++) wp[8 * r] = dc;
      continue;
    }
    long z2 = (long)ip[16] * qp[16], z3 = (long)ip[48] * qp[48];
    long z1 = (z2 + z3) * JFIX_0_541196100;
    long tmp2 = z1 + z3 * (-JFIX_1_847759065);
    long tmp3 = z1 + z2 * JFIX_0_765366865;
    z2 = (long)ip[0] * qp[0];
    z3 = (long)ip[32] * qp[32];
    long tmp0 = (z2 + z3) * (1L << CONST_BITS), tmp1 = (z2 - z3) * (1L << CONST_BITS);
    long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = (long)ip[56] * qp[56];
    tmp1 = (long)ip[40] * qp[40];
    tmp2 = (long)ip[24] * qp[24];
    tmp3 = (long)ip[8] * qp[8];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    long z5 = (z3 + z4) * JFIX_1_175875602;
    tmp0 *= JFIX_0_298631336;
    tmp1 *= JFIX_2_053119869;
    tmp2 *= JFIX_3_072711026;
    tmp3 *= JFIX_1_501321110;
    z1 *= -JFIX_0_899976223;
    z2 *= -JFIX_2_562915447;
    z3 *= -JFIX_1_961570560;
    z4 *= -JFIX_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    wp[0] = jdescale(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
    wp[56] = jdescale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
    wp[8] = jdescale(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
    wp[48] = jdescale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
    wp[16] = jdescale(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
    wp[40] = jdescale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
    wp[24] = jdescale(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
    wp[32] = jdescale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
  }
  for (int r = 0; r < 8; r++) {
    const long* wp = ws + 8 * r;
    uint8_t* op = out + (size_t)r * (size_t)stride;
    if (wp[1] == 0 && wp[2] == 0 && wp[3] == 0 && wp[4] == 0 && wp[5] == 0 && wp[6] == 0 && wp[7] == 0) {
      uint8_t dc = clamp8((int)jdescale(wp[0], PASS1_BITS + 3) + 128);
      for (int c = 0; c < 8; c++) op[c] = dc;
      continue;
    }
    long z2 = wp[2], z3 = wp[6];
    long z1 = (z2 + z3) * JFIX_0_541196100;
    long tmp2 = z1 + z3 * (-JFIX_1_847759065);
    long tmp3 = z1 + z2 * JFIX_0_765366865;
    long tmp0 = (wp[0] + wp[4]) * (1L << CONST_BITS), tmp1 = (wp[0] - wp[4]) * (1L << CONST_BITS);
    long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = wp[7];
    tmp1 = wp[5];
    tmp2 = wp[3];
    tmp3 = wp[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    long z5 = (z3 + z4) * JFIX_1_175875602;
    tmp0 *= JFIX_0_298631336;
    tmp1 *= JFIX_2_053119869;
    tmp2 *= JFIX_3_072711026;
    tmp3 *= JFIX_1_501321110;
    z1 *= -JFIX_0_899976223;
    z2 *= -JFIX_2_562915447;
    z3 *= -JFIX_1_961570560;
    z4 *= -JFIX_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    const int S = CONST_BITS + PASS1_BITS + 3;
    op[0] = clamp8((int)jdescale(tmp10 + tmp3, S) + 128);
    op[7] = clamp8((int)jdescale(tmp10 - tmp3, S) + 128);
    op[1] = clamp8((int)jdescale(tmp11 + tmp2, S) + 128);
    op[6] = clamp8((int)jdescale(tmp11 - tmp2, S) + 128);
    op[2] = clamp8((int)jdescale(tmp12 + tmp1, S) + 128);
    op[5] = clamp8((int)jdescale(tmp12 - tmp1, S) + 128);
    op[3] = clamp8((int)jdescale(tmp13 + tmp0, S) + 128);
    op[4] = clamp8((int)jdescale(tmp13 - tmp0, S) + 128);
  }
}

// component plane -> full-resolution plane (W x H). 2:1 horizontal and 2:1 both ways use the triangle filter.
void upsample(const JComp& c, int hmax, int vmax, int W, int H, std::vector<uint8_t>& out) {
  out.assign((size_t)W * (size_t)H, 0);
  const int stride = c.blocks_w * 8;
  const int hx = hmax / c.h, vx = vmax / c.v;
  const uint8_t* in = c.plane.data();
  const int cw = c.width, ch = c.height;
  if (hx == 1 && vx == 1) {
    for (int y = 0; y < H; y++) memcpy(&out[(size_t)y * W], in + (size_t)y * stride, (size_t)W);
    return;
  }
  if (hx == 2 && vx == 1 && hmax % c.h == 0 && vmax % c.v == 0) {
    std::vector<uint8_t> row((size_t)cw * 2 + 2);
    for (int y = 0; y < H; y++) {
      const uint8_t* ip = in + (size_t)y * stride;
      if (cw == 1) {
        row[0] = row[1] = ip[0];
      } else {
        row[0] = ip[0];
        row[1] = (uint8_t)((ip[0] * 3 + ip[1] + 2) >> 2);
        for (int x = 1; x < cw - 1; x++) {
          const int v = ip[x] * 3;
          row[2 * x] = (uint8_t)((v + ip[x - 1] + 1) >> 2);
          row[2 * x + 1] = (uint8_t)((v + ip[x + 1] + 2) >> 2);
        }
        row[2 * (cw - 1)] = (uint8_t)((ip[cw - 1] * 3 + ip[cw - 2] + 1) >> 2);
        row[2 * (cw - 1) + 1] = ip[cw - 1];
      }
      memcpy(&out[(size_t)y * W], row.data(), (size_t)W);
    }
    return;
  }
  if (hx == 2 && vx == 2 && hmax % c.h == 0 && vmax % c.v == 0) {
    std::vector<int> sum((size_t)cw);
    std::vector<uint8_t> row((size_t)cw * 2 + 2);
    for (int y = 0; y < H; y++) {
      const int iy = y >> 1;
      int ny = (y & 1) ? iy + 1 : iy - 1;  // the nearer of the two neighbouring input rows
      if (ny < 0) ny = 0;
      if (ny > ch - 1) ny = ch - 1;
      const uint8_t* i0 = in + (size_t)iy * stride;
      const uint8_t* i1 = in + (size_t)ny * stride;
      for (int x = 0; x < cw; x++) sum[(size_t)x] = i0[x] * 3 + i1[x];
      if (cw == 1) {
        row[0] = (uint8_t)((sum[0] * 4 + 8) >> 4);
        row[1] = (uint8_t)((sum[0] * 4 + 7) >> 4);
      } else {
        row[0] = (uint8_t)((sum[0] * 4 + 8) >> 4);
        row[1] = (uint8_t)((sum[0] * 3 + sum[1] + 7) >> 4);
        for (int x = 1; x < cw - 1; x++) {
          row[2 * x] = (uint8_t)((sum[(size_t)x] * 3 + sum[(size_t)x - 1] + 8) >> 4);
          row[2 * x + 1] = (uint8_t)((sum[(size_t)x] * 3 + sum[(size_t)x + 1] + 7) >> 4);
        }
        row[2 * (cw - 1)] = (uint8_t)((sum[(size_t)cw - 1] * 3 + sum[(size_t)cw - 2] + 8) >> 4);
        row[2 * (cw - 1) + 1] = (uint8_t)((sum[(size_t)cw - 1] * 4 + 7) >> 4);
      }
      memcpy(&out[(size_t)y * W], row.data(), (size_t)W);
    }
    return;
  }
  // any other ratio: pixel replication (sampling factors that do not divide the maximum are refused earlier)
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) out[(size_t)y * W + x] = in[(size_t)(y * c.v / vmax) * stride + (size_t)(x * c.h / hmax)];
}

int decode_jpeg(const uint8_t* data, size_t size, mt_image* out) {
  uint16_t qt[4][64];
  bool qt_present[4] = {false, false, false, false};
  JHuff dc[4], ac[4];
  for (int i = 0; i < 4; i++) dc[i].present = ac[i].present = false;
  std::vector<JComp> comps;
  int W = 0, H = 0, hmax = 1, vmax = 1, restart = 0;
  bool have_sof = false, adobe = false, jfif = false, any_scan = false, progressive = false;
  int adobe_transform = -1;
  size_t pos = 2;
  for (;;) {
    // next marker
    while (pos < size && data[pos] != 0xff) pos++;
    while (pos < size && data[pos] == 0xff) pos++;
    if (pos >= size) break;
    const int m = data[pos++];
    if (m == 0xd9) break;                                   // EOI
    if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;    // TEM, stray RSTn
    if (pos + 2 > size) return fail(MT_ERR_CORRUPT, "jpeg: truncated segment");
    const size_t len = ((size_t)data[pos] << 8) | data[pos + 1];
    if (len < 2 || pos + len > size) return fail(MT_ERR_CORRUPT, "jpeg: bad segment length");
    const uint8_t* s = data + pos + 2;
    const size_t sl = len - 2;
    if (m == 0xdb) {  // DQT
      size_t o = 0;
      while (o < sl) {
        const int pq = s[o] >> 4, tq = s[o] & 15;
        o++;
        if (tq > 3 || pq > 1 || o + (size_t)(pq ? 128 : 64) > sl) return fail(MT_ERR_CORRUPT, "jpeg: bad DQT");
        for (int i = 0; i < 64; i++) {
          qt[tq][kZigzag[i]] = pq ? (uint16_t)((s[o] << 8) | s[o + 1]) : s[o];
          o += pq ? 2 : 1;
        }
        qt_present[tq] = true;
      }
    } else if (m == 0xc4) {  // DHT
      size_t o = 0;
      while (o < sl) {
        if (o + 17 > sl) return fail(MT_ERR_CORRUPT, "jpeg: bad DHT");
        const int tc = s[o] >> 4, th = s[o] & 15;
        if (tc > 1 || th > 3) return fail(MT_ERR_CORRUPT, "jpeg: bad DHT");
        JHuff& h = tc ? ac[th] : dc[th];
        int total = 0;
        h.bits[0] = 0;
        for (int i = 1; i <= 16; i++) total += (h.bits[i] = s[o + (size_t)i]);
        o += 17;
        if (total > 256 || o + (size_t)total > sl) return fail(MT_ERR_CORRUPT, "jpeg: bad DHT");
        memcpy(h.vals, s + o, (size_t)total);
        o += (size_t)total;
        jhuff_prepare(h);
        h.present = true;
      }
    } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {  // SOF0 / SOF1 / SOF2 (progressive)
      if (have_sof || sl < 6) return fail(MT_ERR_CORRUPT, "jpeg: bad SOF");
      progressive = m == 0xc2;
      if (s[0] != 8) return fail(MT_ERR_UNSUPPORTED, "jpeg: only 8-bit precision");
      H = (s[1] << 8) | s[2];
      W = (s[3] << 8) | s[4];
      const int nc = s[5];
      if (W == 0 || H == 0) return fail(MT_ERR_UNSUPPORTED, "jpeg: DNL-defined height not supported");
      if (W > 32768 || H > 32768) return fail(MT_ERR_UNSUPPORTED, "jpeg: size out of range");
      if ((nc != 1 && nc != 3) || sl < 6 + (size_t)nc * 3) return fail(MT_ERR_UNSUPPORTED, "jpeg: 1 or 3 components only (CMYK/YCCK refused)");
      comps.resize((size_t)nc);
      for (int i = 0; i < nc; i++) {
        JComp& c = comps[(size_t)i];
        c.id = s[6 + 3 * i];
        c.h = s[7 + 3 * i] >> 4;
        c.v = s[7 + 3 * i] & 15;
        c.tq = s[8 + 3 * i];
        if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail(MT_ERR_CORRUPT, "jpeg: bad component");
        if (c.h > hmax) hmax = c.h;
        if (c.v > vmax) vmax = c.v;
      }
      if (nc == 1) comps[0].h = comps[0].v = hmax = vmax = 1;  // a single component is never interleaved
      for (JComp& c : comps)
        if (hmax % c.h || vmax % c.v) return fail(MT_ERR_UNSUPPORTED, "jpeg: fractional sampling ratio");
      const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
      for (JComp& c : comps) {
        c.blocks_w = mcux * c.h;
        c.blocks_h = mcuy * c.v;
        c.width = (W * c.h + hmax - 1) / hmax;
        c.height = (H * c.v + vmax - 1) / vmax;
        c.coef.assign((size_t)c.blocks_w * (size_t)c.blocks_h * 64, 0);
      }
      have_sof = true;
    } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {
      return fail(MT_ERR_UNSUPPORTED, "jpeg: lossless / hierarchical / arithmetic coding not supported");
    } else if (m == 0xdd) {  // DRI
      if (sl < 2) return fail(MT_ERR_CORRUPT, "jpeg: bad DRI");
      restart = (s[0] << 8) | s[1];
    } else if (m == 0xe0) {
      if (sl >= 5 && !memcmp(s, "JFIF", 5)) jfif = true;
    } else if (m == 0xee) {
      if (sl >= 12 && !memcmp(s, "Adobe", 5)) {
        adobe = true;
        adobe_transform = s[11];
      }
    } else if (m == 0xda) {  // SOS
      if (!have_sof || sl < 1) return fail(MT_ERR_CORRUPT, "jpeg: SOS before SOF");
      const int ns = s[0];
      if (ns < 1 || ns > (int)comps.size() || sl < 1 + (size_t)ns * 2 + 3) return fail(MT_ERR_CORRUPT, "jpeg: bad SOS");
      int scan[4];
      for (int i = 0; i < ns; i++) {
        int ci = -1;
        for (size_t k = 0; k < comps.size(); k++)
          if (comps[k].id == s[1 + 2 * i]) ci = (int)k;
        if (ci < 0) return fail(MT_ERR_CORRUPT, "jpeg: unknown scan component");
        comps[(size_t)ci].td = s[2 + 2 * i] >> 4;
        comps[(size_t)ci].ta = s[2 + 2 * i] & 15;
        if (comps[(size_t)ci].td > 3 || comps[(size_t)ci].ta > 3) return fail(MT_ERR_CORRUPT, "jpeg: bad table selector");
        scan[i] = ci;
        comps[(size_t)ci].pred = 0;
      }
      const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
      if (!progressive) {
        if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) return fail(MT_ERR_CORRUPT, "jpeg: spectral selection in a sequential scan");
      } else {
        if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah != 0 && Ah != Al + 1))
          return fail(MT_ERR_CORRUPT, "jpeg: bad progressive scan parameters");
      }
      const bool need_dc = !progressive || (Ss == 0 && Ah == 0), need_ac = !progressive || Ss > 0;
      for (int i = 0; i < ns; i++) {
        const JComp& c = comps[(size_t)scan[i]];
        if ((need_dc && !dc[c.td].present) || (need_ac && !ac[c.ta].present))
          return fail(MT_ERR_CORRUPT, "jpeg: scan refers to a missing Huffman table");
      }
      int eobrun = 0;
      JBits br = {data, size, pos + len, 0, 0, false};
      int units_x, units_y;
      if (ns == 1) {
        const JComp& c = comps[(size_t)scan[0]];
        units_x = (c.width + 7) / 8;
        units_y = (c.height + 7) / 8;
      } else {
        units_x = (W + 8 * hmax - 1) / (8 * hmax);
        units_y = (H + 8 * vmax - 1) / (8 * vmax);
      }
      int to_restart = restart, next_rst = 0;
      for (int uy = 0; uy < units_y; uy++) {
        for (int ux = 0; ux < units_x; ux++) {
          if (restart && to_restart == 0) {
            // byte-align, expect RSTn
            br.cnt = 0;
            br.hit_marker = false;
            size_t q = br.pos;
            while (q + 1 < size && !(data[q] == 0xff && data[q + 1] >= 0xd0 && data[q + 1] <= 0xd7)) {
              if (data[q] == 0xff && data[q + 1] != 0x00 && data[q + 1] != 0xff) break;
              q++;
            }
            if (q + 1 >= size || data[q] != 0xff || data[q + 1] != (uint8_t)(0xd0 + next_rst))
              return fail(MT_ERR_CORRUPT, "jpeg: missing restart marker");
            br.pos = q + 2;
            next_rst = (next_rst + 1) & 7;
            to_restart = restart;
            eobrun = 0;
            for (int i = 0; i < ns; i++) comps[(size_t)scan[i]].pred = 0;
          }
          for (int i = 0; i < ns; i++) {
            JComp& c = comps[(size_t)scan[i]];
            const int bh = ns == 1 ? 1 : c.h, bv = ns == 1 ? 1 : c.v;
            for (int by = 0; by < bv; by++) {
              for (int bx = 0; bx < bh; bx++) {
                const int bxx = ns == 1 ? ux : ux * c.h + bx, byy = ns == 1 ? uy : uy * c.v + by;
                int16_t* blk = &c.coef[((size_t)byy * (size_t)c.blocks_w + (size_t)bxx) * 64];
                if (progressive) {  // T.81 Annex G: one band / one bit plane per scan, coefficients persist across scans
                  if (Ss == 0) {
                    if (Ah == 0) {
                      int t = jhuff_decode(br, dc[c.td]);
                      if (t < 0 || t > 11) return fail(MT_ERR_CORRUPT, "jpeg: bad DC code");
                      c.pred += t ? jextend(br.receive(t), t) : 0;
                      blk[0] = (int16_t)(c.pred * (1 << Al));
                    } else if (br.bit()) {
                      blk[0] = (int16_t)(blk[0] | (1 << Al));
                    }
                  } else if (Ah == 0) {
                    if (eobrun > 0) {
                      eobrun--;
                    } else {
                      for (int k = Ss; k <= Se; k++) {
                        int rs = jhuff_decode(br, ac[c.ta]);
                        if (rs < 0) return fail(MT_ERR_CORRUPT, "jpeg: bad AC code");
                        const int rr = rs >> 4, ss = rs & 15;
                        if (ss) {
                          k += rr;
                          if (k > Se) return fail(MT_ERR_CORRUPT, "jpeg: AC run past the band");
                          blk[kZigzag[k]] = (int16_t)(jextend(br.receive(ss), ss) * (1 << Al));
                        } else if (rr == 15) {
                          k += 15;
                        } else {
                          eobrun = 1 << rr;
                          if (rr) eobrun += br.receive(rr);
                          eobrun--;
                          break;
                        }
                      }
                    }
                  } else {
                    const int p1 = 1 << Al, m1 = -(1 << Al);
                    int k = Ss;
                    if (eobrun == 0) {
                      for (; k <= Se; k++) {
                        int rs = jhuff_decode(br, ac[c.ta]);
                        if (rs < 0) return fail(MT_ERR_CORRUPT, "jpeg: bad AC code");
                        int rr = rs >> 4, ss = rs & 15;
                        if (ss) {
                          if (ss != 1) return fail(MT_ERR_CORRUPT, "jpeg: bad refinement code");
                          ss = br.bit() ? p1 : m1;
                        } else if (rr != 15) {
                          eobrun = 1 << rr;
                          if (rr) eobrun += br.receive(rr);
                          break;
                        }
                        do {
                          int16_t* co = &blk[kZigzag[k]];
                          if (*co != 0) {
                            if (br.bit() && (*co & p1) == 0) *co = (int16_t)(*co + (*co >= 0 ? p1 : m1));
                          } else if (--rr < 0) {
                            break;
                          }
                          k++;
                        } while (k <= Se);
                        if (ss) {
                          if (k > Se) return fail(MT_ERR_CORRUPT, "jpeg: refinement past the band");
                          blk[kZigzag[k]] = (int16_t)ss;
                        }
                      }
                    }
                    if (eobrun > 0) {
                      for (; k <= Se; k++) {
                        int16_t* co = &blk[kZigzag[k]];
                        if (*co != 0 && br.bit() && (*co & p1) == 0) *co = (int16_t)(*co + (*co >= 0 ? p1 : m1));
                      }
                      eobrun--;
                    }
                  }
                  continue;
                }
                int t = jhuff_decode(br, dc[c.td]);
                if (t < 0 || t > 11) return fail(MT_ERR_CORRUPT, "jpeg: bad DC code");
                int diff = t ? jextend(br.receive(t), t) : 0;
                c.pred += diff;
                blk[0] = (int16_t)c.pred;
                for (int k = 1; k < 64;) {
                  int rs = jhuff_decode(br, ac[c.ta]);
                  if (rs < 0) return fail(MT_ERR_CORRUPT, "jpeg: bad AC code");
                  const int rr = rs >> 4, ss = rs & 15;
                  if (ss == 0) {
                    if (rr == 15) {
                      k += 16;
                      continue;
                    }
                    break;  // EOB
                  }
                  k += rr;
                  if (k > 63) return fail(MT_ERR_CORRUPT, "jpeg: AC run past the block");
                  blk[kZigzag[k]] = (int16_t)jextend(br.receive(ss), ss);
                  k++;
                }
              }
            }
          }
          if (restart) to_restart--;
        }
      }
      any_scan = true;
      // continue after the entropy-coded data
      size_t q = br.pos;
      while (q + 1 < size && !(data[q] == 0xff && data[q + 1] != 0x00 && !(data[q + 1] >= 0xd0 && data[q + 1] <= 0xd7) && data[q + 1] != 0xff)) q++;
      pos = q;
      continue;
    }
    pos += len;
  }
  if (!have_sof || !any_scan) return fail(MT_ERR_CORRUPT, "jpeg: no image data");
  for (JComp& c : comps) {
    if (!qt_present[c.tq]) return fail(MT_ERR_CORRUPT, "jpeg: missing quantisation table");
    c.plane.assign((size_t)c.blocks_w * 8 * (size_t)c.blocks_h * 8, 0);
    const int stride = c.blocks_w * 8;
    for (int by = 0; by < c.blocks_h; by++)
      for (int bx = 0; bx < c.blocks_w; bx++)
        idct_islow(&c.coef[((size_t)by * (size_t)c.blocks_w + (size_t)bx) * 64], qt[c.tq],
                   &c.plane[(size_t)by * 8 * (size_t)stride + (size_t)bx * 8], stride);
    std::vector<int16_t>().swap(c.coef);
  }
  uint8_t* rgba = (uint8_t*)malloc((size_t)W * (size_t)H * 4);
  if (!rgba) return fail(MT_ERR_MEMORY, "jpeg: out of memory");
  std::vector<uint8_t> full[3];
  for (size_t i = 0; i < comps.size(); i++) upsample(comps[i], hmax, vmax, W, H, full[i]);
  const size_t npx = (size_t)W * (size_t)H;
  if (comps.size() == 1) {
    for (size_t i = 0; i < npx; i++) {
      rgba[4 * i] = rgba[4 * i + 1] = rgba[4 * i + 2] = full[0][i];
      rgba[4 * i + 3] = 255;
    }
  } else {
    // JFIF says YCbCr; Adobe APP14 transform 0 means the three components already are RGB; without either marker
    // the component ids 'R','G','B' mean RGB (the conventions the common decoders follow)
    bool ycc = true;
    if (adobe) ycc = adobe_transform != 0;
    else if (!jfif && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') ycc = false;
    if (ycc) {
      static int cr_r[256], cb_b[256];
      static long cr_g[256], cb_g[256];
      static bool ready = false;
      if (!ready) {
        for (int i = 0; i < 256; i++) {
          const long x = i - 128;
          cr_r[i] = (int)((91881L * x + 32768L) >> 16);
          cb_b[i] = (int)((116130L * x + 32768L) >> 16);
          cr_g[i] = -46802L * x;
          cb_g[i] = -22554L * x + 32768L;
        }
        ready = true;
      }
      for (size_t i = 0; i < npx; i++) {
        const int y = full[0][i], cb = full[1][i], cr = full[2][i];
        rgba[4 * i] = clamp8(y + cr_r[cr]);
        rgba[4 * i + 1] = clamp8(y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
        rgba[4 * i + 2] = clamp8(y + cb_b[cb]);
        rgba[4 * i + 3] = 255;
      }
    } else {
      for (size_t i = 0; i < npx; i++) {
        rgba[4 * i] = full[0][i];
        rgba[4 * i + 1] = full[1][i];
        rgba[4 * i + 2] = full[2][i];
        rgba[4 * i + 3] = 255;
      }
    }
  }
  out->width = (uint32_t)W;
  out->height = (uint32_t)H;
  out->rgba = rgba;
  return MT_OK;
}

}  // namespace

extern "C" {

int mt_probe(const uint8_t* data, size_t size) {
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (data && size >= 8 && !memcmp(data, png_sig, 8)) return MT_KIND_PNG;
  if (data && size >= 3 && data[0] == 0xff && data[1] == 0xd8 && data[2] == 0xff) return MT_KIND_JPEG;
  return MT_KIND_UNKNOWN;
}

int mt_decode(const uint8_t* data, size_t size, mt_image* out) {
  if (!out) return fail(MT_ERR_FORMAT, "null output");
  out->width = out->height = 0;
  out->rgba = nullptr;
  try {
    switch (mt_probe(data, size)) {
      case MT_KIND_PNG: return decode_png(data, size, out);
      case MT_KIND_JPEG: return decode_jpeg(data, size, out);
      default: return fail(MT_ERR_FORMAT, "not a PNG or JPEG image");
    }
  } catch (const std::bad_alloc&) {
    return fail(MT_ERR_MEMORY, "out of memory");
  }
}

void mt_free(mt_image* img) {
  if (img && img->rgba) {
    free(img->rgba);
    img->rgba = nullptr;
  }
}

const char* mt_last_error(void) { return g_error.c_str(); }

long mt_inflate(const uint8_t* data, size_t size, uint8_t* out, size_t cap) {
  if (!data || (!out && cap)) return fail(MT_ERR_FORMAT, "null argument");
  std::vector<uint8_t> buf;
  try {
    int r = inflate_zlib(data, size, buf, cap);
    if (r == -2) return fail(MT_ERR_MEMORY, "inflate: output larger than the buffer");
    if (r) return fail(MT_ERR_CORRUPT, "inflate: corrupt stream");
  } catch (const std::bad_alloc&) {
    return fail(MT_ERR_MEMORY, "out of memory");
  }
  if (!buf.empty()) memcpy(out, buf.data(), buf.size());
  return (long)buf.size();
}
}
