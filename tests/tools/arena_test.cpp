// CPU unit test of tl_forward's arena allocator (treelearn_amd/csrc/tl_arena.h): random take / give sequences against a brute-force model.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -I treelearn_amd/csrc tests/tools/arena_test.cpp -o arena_test && ./arena_test
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <random>
#include "tl_arena.h"

static int fail(const char* what, int seed, int step) { printf("FAIL %s (seed %d, step %d)\n", what, seed, step); return 1; }

int main() {
  for (int seed = 0; seed < 200; ++seed) {
    std::mt19937 rng(seed);
    Arena a, dry;                                           // the same sequence on a dry copy must give the same offsets and peak
    a.dry = false; a.cap = int64_t(1) << 40;
    std::map<int64_t, int64_t> live;                        // offset -> rounded bytes
    int64_t live_bytes = 0, max_live = 0;
    const int64_t sizes[6] = {100, 4096, 118 << 20, 141 << 20, 237 << 20, 1};
    for (int step = 0; step < 400; ++step) {
      const bool take = live.empty() || (rng() % 100) < 55;
      if (take) {
        const int64_t b = (rng() % 3 == 0) ? (int64_t)(rng() % 1000000 + 1) : sizes[rng() % 6];
        const int64_t rb = (b + 255) & ~int64_t(255);
        const int64_t o = a.take(b), od = dry.take(b);
        if (o != od) return fail("dry run diverges", seed, step);
        if (o % 256) return fail("alignment", seed, step);
        auto nx = live.lower_bound(o);                      // no overlap with any live block
        if (nx != live.end() && nx->first < o + rb) return fail("overlap with the next block", seed, step);
        if (nx != live.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second > o) return fail("overlap with the previous block", seed, step); }
        if (o + rb > a.peak) return fail("block beyond the peak", seed, step);
        live[o] = rb; live_bytes += rb;
        if (live_bytes > max_live) max_live = live_bytes;
      } else {
        auto it = live.begin(); std::advance(it, rng() % live.size());
        a.give(it->first, it->second); dry.give(it->first, it->second);
        live_bytes -= it->second; live.erase(it);
      }
      for (size_t i = 0; i + 1 < a.free_.size(); ++i)       // the free list stays sorted and fully coalesced
        if (a.free_[i].off + a.free_[i].bytes >= a.free_[i + 1].off) return fail("free list not sorted / coalesced", seed, step);
      for (const auto& f : a.free_) {                       // ... and never overlaps a live block
        auto nx = live.lower_bound(f.off);
        if (nx != live.end() && nx->first < f.off + f.bytes) return fail("free block overlaps a live one", seed, step);
      }
    }
    if (a.peak != dry.peak || a.peak < max_live) return fail("peak", seed, 0);
    for (auto& kv : live) a.give(kv.first, kv.second);      // everything returned: one free block [0, cur) or an empty arena
    if (!(a.free_.size() == 1 && a.free_[0].off == 0 && a.free_[0].bytes == a.cur) && !(a.free_.empty() && a.cur == 0)) return fail("not fully coalesced at the end", seed, 0);
  }
  printf("arena_test OK\n");
  return 0;
}
