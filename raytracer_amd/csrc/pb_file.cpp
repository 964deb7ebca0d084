// The on-disk form of a segment set: what BuildTree holds (CastPhotons.h:83, :808) plus the root box that
// CastPhotonBeams returns (CastPhotons.h:384-389), as pb_segment records.  Host code only.
//
//   offset  0  char[8]   "PBSEGS\0\1"
//           8  uint32    format version (1)
//          12  uint32    record size in bytes (112 = sizeof(pb_segment))
//          16  uint64    number of records n
//          24  float[3]  root box, min_b of HitBoundingBox (un-flipped)
//          36  float[3]  root box, max_b
//          48  16 bytes  reserved, zero
//          64  n records, little-endian, in the order they were uploaded or emitted
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/photon_beams.h"

void pb_set_error(const std::string& msg);

namespace {
const char MAGIC[8] = {'P', 'B', 'S', 'E', 'G', 'S', '\0', '\1'};
struct Header {
  char magic[8];
  uint32_t version, record_size;
  uint64_t n;
  float root_lo[3], root_hi[3];
  unsigned char reserved[16];
};
static_assert(sizeof(Header) == 64, "segment file header");
static_assert(sizeof(pb_segment) == 112, "segment record");
}  // namespace

extern "C" {

int pb_write_segments_file(const char* path, const pb_segment* segs, int n, const float root_lo[3], const float root_hi[3]) {
  if (!path || (!segs && n > 0) || n < 0 || !root_lo || !root_hi) { pb_set_error("pb_write_segments_file: bad argument"); return PB_ERR_ARG; }
  Header h;
  std::memset(&h, 0, sizeof(h));
  std::memcpy(h.magic, MAGIC, 8);
  h.version = 1; h.record_size = (uint32_t)sizeof(pb_segment); h.n = (uint64_t)n;
  for (int k = 0; k < 3; ++k) { h.root_lo[k] = root_lo[k]; h.root_hi[k] = root_hi[k]; }
  FILE* f = std::fopen(path, "wb");
  if (!f) { pb_set_error(std::string("pb_write_segments_file: cannot open ") + path); return PB_ERR_ARG; }
  bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1;
  if (ok && n > 0) ok = std::fwrite(segs, sizeof(pb_segment), (size_t)n, f) == (size_t)n;
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) { pb_set_error(std::string("pb_write_segments_file: short write to ") + path); return PB_ERR_DEVICE; }
  return PB_OK;
}

int pb_read_segments_file(const char* path, pb_segment* out, int capacity, int* n_out, float root_lo[3], float root_hi[3]) {
  if (!path || !n_out) { pb_set_error("pb_read_segments_file: bad argument"); return PB_ERR_ARG; }
  FILE* f = std::fopen(path, "rb");
  if (!f) { pb_set_error(std::string("pb_read_segments_file: cannot open ") + path); return PB_ERR_ARG; }
  Header h;
  if (std::fread(&h, sizeof(h), 1, f) != 1 || std::memcmp(h.magic, MAGIC, 8) != 0 || h.version != 1 ||
      h.record_size != sizeof(pb_segment) || h.n > 0x7fffffffull) {
    std::fclose(f);
    pb_set_error(std::string("pb_read_segments_file: ") + path + " is not a version-1 segment file of 112-byte records");
    return PB_ERR_ARG;
  }
  *n_out = (int)h.n;
  if (root_lo) for (int k = 0; k < 3; ++k) root_lo[k] = h.root_lo[k];
  if (root_hi) for (int k = 0; k < 3; ++k) root_hi[k] = h.root_hi[k];
  if (!out) { std::fclose(f); return PB_OK; }            // a query for the count
  if ((uint64_t)capacity < h.n) { std::fclose(f); pb_set_error("pb_read_segments_file: output array too small"); return PB_ERR_CAPACITY; }
  const bool ok = h.n == 0 || std::fread(out, sizeof(pb_segment), (size_t)h.n, f) == (size_t)h.n;
  std::fclose(f);
  if (!ok) { pb_set_error(std::string("pb_read_segments_file: ") + path + " is shorter than its header says"); return PB_ERR_ARG; }
  return PB_OK;
}

}  // extern "C"
