// ovl_pool.h -- caching device allocator of the overlap library.  hipMalloc / hipFree of multi-GB buffers cost
// milliseconds each (more when the HBM is nearly full), and every index build / map / sort call asks for the same
// sizes again: freed blocks are kept (rounded up to 1/8-octave size classes) and handed back to the next request.
#pragma once

#include <cstddef>

namespace ndovl {
void *pool_alloc(size_t bytes);   // throws std::runtime_error on failure
void pool_free(void *p);
size_t pool_trim();               // idle slabs back to the driver; returns the bytes released
size_t pool_cached_bytes();
}
