// Arena: the device-memory allocator of tl_forward (csrc/tl_exec.hip) -- a best-fit free list with coalescing over ONE caller-provided block.
// Host-only, no HIP: offsets are plain integers, so the same code plans a forward (dry = true: nothing is enqueued, `peak` is the exact
// requirement) and then carves the caller's arena.  Unit-tested on the CPU (tests/test_host_cpu.py compiles tests/tools/arena_test.cpp with
// g++ -fsanitize=address,undefined).
#pragma once
#include <stdint.h>
#include <vector>

struct Arena {                                   // best-fit free list with coalescing over the caller's block; dry = measure only
  char* base = nullptr;
  int64_t cap = 0, cur = 0, peak = 0;
  bool dry = true;
  struct Slot { int64_t off, bytes; };
  std::vector<Slot> free_;                       // sorted by offset, neighbours merged
  int64_t take(int64_t bytes) {
    bytes = (bytes + 255) & ~int64_t(255);
    int best = -1;
    for (size_t i = 0; i < free_.size(); ++i)
      if (free_[i].bytes >= bytes && (best < 0 || free_[i].bytes < free_[best].bytes)) best = (int)i;
    if (best >= 0) {
      const int64_t o = free_[best].off;
      if (free_[best].bytes == bytes) free_.erase(free_.begin() + best);
      else { free_[best].off += bytes; free_[best].bytes -= bytes; }
      return o;
    }
    int64_t o = cur;
    if (!free_.empty() && free_.back().off + free_.back().bytes == cur) {      // a free block at the very end grows instead of being skipped
      o = free_.back().off;
      free_.pop_back();
    }
    cur = o + bytes;
    if (cur > peak) peak = cur;
    return o;
  }
  void give(int64_t off, int64_t bytes) {
    bytes = (bytes + 255) & ~int64_t(255);
    size_t i = 0;
    while (i < free_.size() && free_[i].off < off) ++i;
    free_.insert(free_.begin() + i, Slot{off, bytes});
    if (i + 1 < free_.size() && free_[i].off + free_[i].bytes == free_[i + 1].off) { free_[i].bytes += free_[i + 1].bytes; free_.erase(free_.begin() + i + 1); }
    if (i > 0 && free_[i - 1].off + free_[i - 1].bytes == free_[i].off) { free_[i - 1].bytes += free_[i].bytes; free_.erase(free_.begin() + i); }
  }
  char* at(int64_t off) const { return base + off; }      // (dry: base = nullptr, the pointer is never used)
};
