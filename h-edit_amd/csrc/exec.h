// Pieces shared by the network executors (unet.hip, vae.hip).
#pragma once
#include <map>

#include "common.h"

namespace {

// first-fit arena over the caller's workspace; "dry" mode only records the peak
struct Arena {
  char* base = nullptr;
  size_t cap = 0, peak = 0;
  bool dry = false;
  std::map<size_t, size_t> used;   // offset -> size
  bool failed = false;

  void* alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    size_t pos = 0;
    for (auto& kv : used) {
      if (kv.first >= pos + bytes) break;
      pos = kv.first + kv.second;
    }
    if (!dry && pos + bytes > cap) { failed = true; return nullptr; }
    used[pos] = bytes;
    if (pos + bytes > peak) peak = pos + bytes;
    return dry ? reinterpret_cast<void*>(pos + 4096) : base + pos;
  }
  void free(void* p) {
    if (!p) return;
    size_t off = dry ? reinterpret_cast<size_t>(p) - 4096 : (size_t)(reinterpret_cast<char*>(p) - base);
    used.erase(off);
  }
};


}  // namespace
