// pinflate.h -- a gzip file inflated by several host threads (pinflate.cpp); the reading side of fastx_reader.cpp.
#pragma once

#include <cstdint>

namespace ndovl {

struct PInflate;
// nullptr: not a regular file that starts with a gzip member, or threads < 2 -- the caller reads it with zlib's gzread
PInflate *pinflate_open(const char *path, int threads);
// as gzread: the next bytes of the concatenated members; < len only at the end of the data; -1: corrupt data (a bad code, a
// member whose CRC-32 or length does not match) once everything in front of the failing round has been handed out
int64_t pinflate_read(PInflate *h, void *buf, uint64_t len);
// rounds run, chunks decoded, chunks accepted (every round accepts its first chunk)
void pinflate_stats(PInflate *h, uint64_t out[3]);
void pinflate_close(PInflate *h);

}  // namespace ndovl
