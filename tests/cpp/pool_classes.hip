// Host-logic check of mcp::DevCache's size classes (mcptam_amd/csrc/ba_pool.h): prints "request class_bytes class" for a sweep of
// request sizes.  No device needed.
#include "../../mcptam_amd/csrc/ba_pool.h"
#include <cstdio>
int main() {
  const size_t reqs[] = {1, 511, 512, 513, 640, 641, 1000, 4096, 4097, 5000, 65536, 65537, 100000, 1u << 20, (1u << 20) + 1, 3000000, 41943040, 41943041, (size_t)5 << 30};
  for (size_t r : reqs) { int c = -1; const size_t b = mcp::DevCache::class_bytes(r, &c); printf("%zu %zu %d\n", r, b, c); }
  return 0;
}
