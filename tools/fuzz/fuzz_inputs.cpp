// mutation fuzzer for mt_decode and ms_world_create_glb (built with -fsanitize=address,undefined)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "mi355tex.h"
#include "mi355scene.h"
static uint64_t s = 88172645463325252ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
static std::vector<uint8_t> load(const char* p) { std::vector<uint8_t> v; FILE* f = fopen(p, "rb"); if (!f) return v; int c; while ((c = fgetc(f)) != EOF) v.push_back((uint8_t)c); fclose(f); return v; }
static void mutate(std::vector<uint8_t>& d) {
  int n = 1 + rnd() % 6;
  for (int i = 0; i < n && !d.empty(); i++) {
    switch (rnd() % 6) {
      case 0: d[rnd() % d.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
      case 1: d[rnd() % d.size()] = (uint8_t)rnd(); break;
      case 2: d.resize(rnd() % (d.size() + 1)); break;
      case 3: { size_t p = rnd() % d.size(); d.insert(d.begin() + p, (uint8_t)rnd()); } break;
      case 4: { size_t p = rnd() % d.size(); size_t l = rnd() % 16; if (p + l < d.size()) d.erase(d.begin() + p, d.begin() + p + l); } break;
      case 5: { size_t p = rnd() % d.size(); const uint8_t v[] = {0, 0xff, 0x7f, 0x80}; d[p] = v[rnd() % 4]; if (p + 1 < d.size()) d[p + 1] = v[rnd() % 4]; } break;
    }
  }
}
int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 2000;
  long ok = 0, bad = 0;
  for (int f = 0; f < 40; f++) {
    char name[64];
    snprintf(name, sizeof name, "img_%02d.bin", f);
    std::vector<uint8_t> seed = load(name);
    if (seed.empty()) continue;
    for (int i = 0; i < iters; i++) {
      std::vector<uint8_t> d = seed;
      mutate(d);
      mt_image im;
      if (mt_decode(d.data(), d.size(), &im) == MT_OK) { ok++; volatile uint8_t x = im.rgba[(size_t)im.width * im.height * 4 - 1]; (void)x; mt_free(&im); } else bad++;
    }
  }
  printf("images: %ld decoded, %ld refused\n", ok, bad);
  ok = bad = 0;
  for (int f = 0; f < 8; f++) {
    char name[64];
    snprintf(name, sizeof name, "glb_%02d.bin", f);
    std::vector<uint8_t> seed = load(name);
    if (seed.empty()) continue;
    for (int i = 0; i < iters / 4; i++) {
      std::vector<uint8_t> d = seed;
      if (i % 2 == 0) {
        mutate(d);
      } else {
        // keep the JSON well-formed: overwrite 1..4 digit runs of the JSON chunk with other digits of the same width
        size_t json_end = d.size();
        if (d.size() > 20 && !memcmp(d.data(), "glTF", 4)) { uint32_t l; memcpy(&l, d.data() + 12, 4); json_end = 20 + (size_t)l < d.size() ? 20 + (size_t)l : d.size(); }
        for (int k = 0, n = 1 + rnd() % 4; k < n; k++) {
          size_t p = 20 + rnd() % (json_end > 21 ? json_end - 21 : 1);
          while (p < json_end && !(d[p] >= '0' && d[p] <= '9')) p++;
          while (p < json_end && d[p] >= '0' && d[p] <= '9') { d[p] = (uint8_t)('0' + rnd() % 10); p++; if (rnd() % 3 == 0) break; }
        }
      }
      ms_world* w = ms_world_create_glb("viewer", nullptr, d.data(), d.size());
      if (w) { if (ms_last_error()[0]) bad++; else ok++; ms_world_update(w, 0.37f); ms_world_update(w, 1.9f); ms_world_destroy(w); }
    }
  }
  printf("glb: %ld loaded, %ld refused\n", ok, bad);
  return 0;
}
