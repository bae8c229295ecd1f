// .npy reader/writer for the embedding index (proqa_npy_* in proqa_hip.h).
//
// The reference writes the index with np.save (/root/reference/retrieval/get_embed.py:139) and
// reads it with np.load (retrieval/eval_retrieval.py:99-100): 2-D C-order arrays of '<f2' (under
// --fp16) or '<f4'.  Files written here are byte-identical to numpy's for the same array
// (format v1.0, dict literal in numpy's key order, header padded with spaces to a multiple of
// 64 bytes and terminated by '\n').
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

namespace proqa {
namespace {

const unsigned char kMagic[6] = {0x93, 'N', 'U', 'M', 'P', 'Y'};

struct File {
  FILE* f = nullptr;
  ~File() {
    if (f) fclose(f);
  }
};

// Writers end with this: the buffered tail is flushed (and, for files several ranks fill, synced) and the close
// result is checked, so that ENOSPC / quota / NFS errors surface as PROQA_EIO like np.save would raise them.
int close_written(File& fh, const char* path, bool sync) {
  FILE* f = fh.f;
  fh.f = nullptr;
  int err = 0;
  if (fflush(f) != 0) err = errno;
  if (!err && sync && fsync(fileno(f)) != 0) err = errno;
  if (fclose(f) != 0 && !err) err = errno;
  if (err) return fail(PROQA_EIO, "%s: flushing the file failed: %s", path, strerror(err));
  return PROQA_OK;
}

bool find_value(const std::string& h, const char* key, size_t* pos) {
  std::string pat = std::string("'") + key + "'";
  size_t p = h.find(pat);
  if (p == std::string::npos) return false;
  p = h.find(':', p + pat.size());
  if (p == std::string::npos) return false;
  ++p;
  while (p < h.size() && h[p] == ' ') ++p;
  *pos = p;
  return true;
}

int parse_header(FILE* f, const char* path, proqa_npy_info* info) {
  unsigned char pre[12];
  if (fread(pre, 1, 10, f) != 10) return fail(PROQA_EFORMAT, "%s: shorter than a .npy preamble", path);
  if (memcmp(pre, kMagic, 6) != 0) return fail(PROQA_EFORMAT, "%s: bad .npy magic", path);
  const int major = pre[6];
  size_t hlen = 0, pre_len = 10;
  if (major == 1) {
    hlen = (size_t)pre[8] | ((size_t)pre[9] << 8);
  } else if (major == 2 || major == 3) {
    if (fread(pre + 10, 1, 2, f) != 2) return fail(PROQA_EFORMAT, "%s: truncated v%d preamble", path, major);
    hlen = (size_t)pre[8] | ((size_t)pre[9] << 8) | ((size_t)pre[10] << 16) | ((size_t)pre[11] << 24);
    pre_len = 12;
  } else {
    return fail(PROQA_EFORMAT, "%s: unsupported .npy version %d.%d", path, major, pre[7]);
  }
  if (hlen == 0 || hlen > (1u << 20)) return fail(PROQA_EFORMAT, "%s: implausible header length %zu", path, hlen);
  std::string h(hlen, '\0');
  if (fread(&h[0], 1, hlen, f) != hlen) return fail(PROQA_EFORMAT, "%s: truncated header", path);

  size_t p = 0;
  if (!find_value(h, "descr", &p) || p >= h.size() || (h[p] != '\'' && h[p] != '"'))
    return fail(PROQA_EFORMAT, "%s: header has no 'descr'", path);
  const size_t q = h.find(h[p], p + 1);
  if (q == std::string::npos) return fail(PROQA_EFORMAT, "%s: malformed 'descr'", path);
  const std::string descr = h.substr(p + 1, q - p - 1);
  if (descr == "<f2" || descr == "=f2" || descr == "|f2") {
    info->dtype = PROQA_F16;
  } else if (descr == "<f4" || descr == "=f4") {
    info->dtype = PROQA_F32;
  } else {
    return fail(PROQA_EFORMAT, "%s: dtype '%s' is not '<f2' or '<f4'", path, descr.c_str());
  }

  if (!find_value(h, "fortran_order", &p)) return fail(PROQA_EFORMAT, "%s: header has no 'fortran_order'", path);
  if (h.compare(p, 5, "False") != 0) return fail(PROQA_EFORMAT, "%s: Fortran-order arrays are not supported", path);

  if (!find_value(h, "shape", &p) || p >= h.size() || h[p] != '(')
    return fail(PROQA_EFORMAT, "%s: header has no 'shape'", path);
  const size_t close = h.find(')', p);
  if (close == std::string::npos) return fail(PROQA_EFORMAT, "%s: malformed 'shape'", path);
  std::vector<long long> dims;
  const char* s = h.c_str() + p + 1;
  const char* end = h.c_str() + close;
  while (s < end) {
    while (s < end && (*s == ' ' || *s == ',')) ++s;
    if (s >= end) break;
    char* nx = nullptr;
    errno = 0;
    long long v = strtoll(s, &nx, 10);
    if (nx == s || errno || v < 0) return fail(PROQA_EFORMAT, "%s: malformed 'shape'", path);
    dims.push_back(v);
    s = nx;
    if (s < end && *s == 'L') ++s;  // python-2 era long suffix
  }
  if (dims.size() != 2) return fail(PROQA_EFORMAT, "%s: expected a 2-D array, got %zu-D", path, dims.size());
  info->rows = dims[0];
  info->cols = dims[1];
  info->data_offset = (int64_t)(pre_len + hlen);
  return PROQA_OK;
}

std::string make_header(int64_t rows, int64_t cols, int dtype) {
  char dict[160];
  snprintf(dict, sizeof dict, "{'descr': '%s', 'fortran_order': False, 'shape': (%lld, %lld), }",
           dtype == PROQA_F16 ? "<f2" : "<f4", (long long)rows, (long long)cols);
  std::string h(dict);
  // numpy pads with spaces so that preamble(10) + header (incl. trailing '\n') is a multiple of 64
  const size_t unpadded = 10 + h.size() + 1;
  const size_t pad = (64 - unpadded % 64) % 64;
  h.append(pad, ' ');
  h.push_back('\n');
  std::string out((const char*)kMagic, 6);
  out.push_back(1);
  out.push_back(0);
  out.push_back((char)(h.size() & 0xff));
  out.push_back((char)((h.size() >> 8) & 0xff));
  out += h;
  return out;
}

size_t elem_size(int dtype) { return dtype == PROQA_F16 ? 2 : 4; }

}  // namespace
}  // namespace proqa

using namespace proqa;

extern "C" {

int proqa_npy_stat(const char* path, proqa_npy_info* info) {
  if (!path || !info) return fail(PROQA_EINVAL, "npy_stat: NULL argument");
  File fh;
  fh.f = fopen(path, "rb");
  if (!fh.f) return fail(PROQA_EIO, "cannot open %s: %s", path, strerror(errno));
  if (int rc = parse_header(fh.f, path, info)) return rc;
  if (fseeko(fh.f, 0, SEEK_END) != 0) return fail(PROQA_EIO, "%s: seek failed", path);
  const long long size = ftello(fh.f);
  const long long need = info->data_offset + info->rows * info->cols * (long long)elem_size(info->dtype);
  if (size < need) return fail(PROQA_EFORMAT, "%s: file has %lld bytes, header promises %lld", path, size, need);
  return PROQA_OK;
}

int proqa_npy_read_rows(const char* path, int64_t row0, int64_t n, void* dst, size_t dst_bytes) {
  if (!path || (!dst && n > 0) || row0 < 0 || n < 0) return fail(PROQA_EINVAL, "npy_read_rows: bad argument");
  proqa_npy_info info;
  File fh;
  fh.f = fopen(path, "rb");
  if (!fh.f) return fail(PROQA_EIO, "cannot open %s: %s", path, strerror(errno));
  if (int rc = parse_header(fh.f, path, &info)) return rc;
  if (row0 + n > info.rows) return fail(PROQA_EINVAL, "%s: rows [%lld,%lld) out of range (%lld rows)", path,
                                        (long long)row0, (long long)(row0 + n), (long long)info.rows);
  const size_t row_bytes = (size_t)info.cols * elem_size(info.dtype);
  const size_t bytes = (size_t)n * row_bytes;
  if (bytes > dst_bytes) return fail(PROQA_EINVAL, "npy_read_rows: destination too small (%zu < %zu)", dst_bytes, bytes);
  if (fseeko(fh.f, (off_t)(info.data_offset + (long long)row0 * (long long)row_bytes), SEEK_SET) != 0)
    return fail(PROQA_EIO, "%s: seek failed", path);
  if (bytes && fread(dst, 1, bytes, fh.f) != bytes) return fail(PROQA_EFORMAT, "%s: truncated data", path);
  return PROQA_OK;
}

int proqa_npy_write(const char* path, const void* data, int64_t rows, int64_t cols, int dtype) {
  if (!path || rows < 0 || cols < 0 || (!data && rows * cols > 0)) return fail(PROQA_EINVAL, "npy_write: bad argument");
  if (dtype != PROQA_F16 && dtype != PROQA_F32) return fail(PROQA_EINVAL, "npy_write: bad dtype %d", dtype);
  File fh;
  fh.f = fopen(path, "wb");
  if (!fh.f) return fail(PROQA_EIO, "cannot create %s: %s", path, strerror(errno));
  const std::string h = make_header(rows, cols, dtype);
  const size_t bytes = (size_t)rows * cols * elem_size(dtype);
  if (fwrite(h.data(), 1, h.size(), fh.f) != h.size() || (bytes && fwrite(data, 1, bytes, fh.f) != bytes))
    return fail(PROQA_EIO, "%s: write failed: %s", path, strerror(errno));
  return close_written(fh, path, false);
}

int proqa_npy_create(const char* path, int64_t rows, int64_t cols, int dtype) {
  if (!path || rows < 0 || cols < 0) return fail(PROQA_EINVAL, "npy_create: bad argument");
  if (dtype != PROQA_F16 && dtype != PROQA_F32) return fail(PROQA_EINVAL, "npy_create: bad dtype %d", dtype);
  File fh;
  fh.f = fopen(path, "wb");
  if (!fh.f) return fail(PROQA_EIO, "cannot create %s: %s", path, strerror(errno));
  const std::string h = make_header(rows, cols, dtype);
  if (fwrite(h.data(), 1, h.size(), fh.f) != h.size()) return fail(PROQA_EIO, "%s: write failed", path);
  const long long total = (long long)h.size() + (long long)rows * cols * (long long)elem_size(dtype);
  if (fflush(fh.f) != 0 || ftruncate(fileno(fh.f), (off_t)total) != 0)
    return fail(PROQA_EIO, "%s: cannot size file to %lld bytes: %s", path, total, strerror(errno));
  return close_written(fh, path, true);
}

int proqa_npy_write_rows(const char* path, int64_t row0, int64_t n, const void* src, int64_t cols, int dtype) {
  if (!path || row0 < 0 || n < 0 || (!src && n > 0)) return fail(PROQA_EINVAL, "npy_write_rows: bad argument");
  proqa_npy_info info;
  File fh;
  fh.f = fopen(path, "r+b");
  if (!fh.f) return fail(PROQA_EIO, "cannot open %s for update: %s", path, strerror(errno));
  if (int rc = parse_header(fh.f, path, &info)) return rc;
  if (cols != info.cols || dtype != info.dtype)
    return fail(PROQA_EINVAL, "%s holds %lld-column rows of dtype %d; the rows to write have %lld columns of dtype %d",
                path, (long long)info.cols, info.dtype, (long long)cols, dtype);
  if (row0 + n > info.rows) return fail(PROQA_EINVAL, "%s: rows [%lld,%lld) out of range (%lld rows)", path,
                                        (long long)row0, (long long)(row0 + n), (long long)info.rows);
  const size_t row_bytes = (size_t)info.cols * elem_size(info.dtype);
  if (fseeko(fh.f, (off_t)(info.data_offset + (long long)row0 * (long long)row_bytes), SEEK_SET) != 0)
    return fail(PROQA_EIO, "%s: seek failed", path);
  const size_t bytes = (size_t)n * row_bytes;
  if (bytes && fwrite(src, 1, bytes, fh.f) != bytes) return fail(PROQA_EIO, "%s: write failed: %s", path, strerror(errno));
  return close_written(fh, path, true);
}

}  // extern "C"
