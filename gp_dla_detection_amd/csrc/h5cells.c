/* h5cells.c -- native reader for the cells of a MATLAB -v7.3 (HDF5) cell array of numeric vectors:
 * the four ragged arrays of preloaded_qsos.mat (preload_qsos.m:64-79), which a sharded run reads
 * quasar by quasar while the GPU sweeps (run_dr12q.py).  MATLAB's `load` is compiled code; the
 * pure-Python reader of hdf5.py spends ~30 us per cell on header parsing under the GIL, which makes
 * reading (not sweeping) the longer leg of a DR12Q shard.  This file restates, for that one case,
 * what hdf5.py does in general: version-1 object headers (with continuation blocks), dataspace,
 * datatype, data-layout versions 1-3 (compact / contiguous / chunked through a version-1 B-tree) and
 * the filter pipeline (deflate, shuffle, fletcher32), with the cells spread over OpenMP threads.
 * The file is the caller's read-only memory map (`file`, `file_len`): no system call per cell.
 *
 * Anything outside that subset -- version-2 headers, other layouts or filters, a cell that is not
 * a vector, an empty cell (MATLAB stores its dimensions, rank 1) -- is REPORTED per cell (count -1)
 * and left to the Python reader; nothing is guessed.  No libhdf5.
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp h5cells.c -lz -o libgpdla_h5cells.so (io.py does it). */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define UNDEF_ADDR 0xFFFFFFFFFFFFFFFFull
#define MAX_RANK 4
#define MAX_FILTERS 4

typedef struct {
  int rank;
  uint64_t dims[MAX_RANK];
  int elem_size;
  int type_class;          /* 0 fixed-point (integer), 1 floating-point */
  int layout;              /* 0 compact, 1 contiguous, 2 chunked */
  uint64_t addr;           /* contiguous: data; chunked: B-tree */
  uint64_t size;           /* contiguous / compact: bytes */
  uint64_t compact_at;     /* compact: absolute file offset of the data inside the header */
  uint32_t cdims[MAX_RANK];
  int nfilters;
  int filter_id[MAX_FILTERS];
  uint32_t filter_cd0[MAX_FILTERS];
} cell_info;

typedef struct {
  const uint8_t *p;
  uint64_t len;
} h5file;
typedef const h5file *fd_t;

/* bounds-checked copy out of the mapped file */
static int read_at(fd_t f, uint64_t off, void *buf, size_t n) {
  if (off > f->len || n > f->len - off) return -1;
  memcpy(buf, f->p + off, n);
  return 0;
}
static uint16_t u16(const uint8_t *p) { return (uint16_t)(p[0] | p[1] << 8); }
static uint32_t u32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint64_t u64(const uint8_t *p) { return (uint64_t)u32(p) | (uint64_t)u32(p + 4) << 32; }

/* One message of a version-1 object header.  Returns 0, or -1 for what this reader does not take. */
static int take_message(cell_info *ci, int type, const uint8_t *d, size_t n, uint64_t abs_at, int *seen) {
  if (type == 0x0001) { /* dataspace */
    if (n < 8) return -1;
    int version = d[0], rank = d[1];
    size_t pos = version == 1 ? 8 : 4;
    if ((version != 1 && version != 2) || rank < 1 || rank > MAX_RANK || n < pos + 8u * (size_t)rank) return -1;
    if (version == 2 && d[3] != 1) return -1; /* simple dataspaces only */
    ci->rank = rank;
    for (int i = 0; i < rank; ++i) ci->dims[i] = u64(d + pos + 8 * i);
    *seen |= 1;
  } else if (type == 0x0003) { /* datatype: fixed-point or floating-point, little-endian */
    if (n < 8) return -1;
    int cls = d[0] & 0x0F;
    if ((cls != 0 && cls != 1) || (d[1] & 1)) return -1;
    ci->type_class = cls;
    ci->elem_size = (int)u32(d + 4);
    *seen |= 2;
  } else if (type == 0x0008 && n >= 8 && (d[0] == 1 || d[0] == 2)) { /* data layout, versions 1 and 2 (HDF5 1.6) */
    int nd = d[1], cls = d[2];
    size_t pos = 8;
    if (nd < 1 || nd > MAX_RANK + 1 || cls > 2) return -1;
    if (cls != 0) {
      if (n < pos + 8) return -1;
      ci->addr = u64(d + pos);
      pos += 8;
    }
    if (n < pos + 4u * (size_t)nd) return -1;
    ci->layout = cls;
    if (cls == 2) {
      if (nd < 2) return -1;
      for (int i = 0; i < nd - 1; ++i) ci->cdims[i] = u32(d + pos + 4 * i);
    } else if (cls == 1) {
      ci->size = ~(uint64_t)0; /* not stored in these versions: the dataspace says how much */
    } else {
      pos += 4u * (size_t)nd;
      if (n < pos + 4) return -1;
      ci->size = u32(d + pos);
      if (n < pos + 4 + ci->size) return -1;
      ci->compact_at = abs_at + pos + 4;
    }
    *seen |= 4;
  } else if (type == 0x0008) { /* data layout, version 3 */
    if (n < 2 || d[0] != 3) return -1;
    ci->layout = d[1];
    if (d[1] == 0) {
      if (n < 4) return -1;
      ci->size = u16(d + 2);
      if (n < 4 + ci->size) return -1;
      ci->compact_at = abs_at + 4;
    } else if (d[1] == 1) {
      if (n < 18) return -1;
      ci->addr = u64(d + 2);
      ci->size = u64(d + 10);
    } else if (d[1] == 2) {
      if (n < 11) return -1;
      int nd = d[2];
      if (nd < 2 || nd > MAX_RANK + 1 || n < 11u + 4u * (size_t)nd) return -1;
      ci->addr = u64(d + 3);
      for (int i = 0; i < nd - 1; ++i) ci->cdims[i] = u32(d + 11 + 4 * i);
    } else {
      return -1;
    }
    *seen |= 4;
  } else if (type == 0x000B) { /* filter pipeline */
    if (n < 2) return -1;
    int version = d[0], nf = d[1];
    if ((version != 1 && version != 2) || nf > MAX_FILTERS) return -1;
    size_t pos = version == 1 ? 8 : 2;
    for (int f = 0; f < nf; ++f) {
      if (pos + 8 > n) return -1;
      int fid = u16(d + pos), nlen = 0, ncd;
      if (version == 1 || fid >= 256) {
        nlen = u16(d + pos + 2);
        ncd = u16(d + pos + 6);
        pos += 8;
      } else {
        ncd = u16(d + pos + 4);
        pos += 6;
      }
      pos += version == 1 ? (size_t)((nlen + 7) & ~7) : (size_t)nlen;
      if (pos + 4u * (size_t)ncd > n) return -1;
      if (fid != 1 && fid != 2 && fid != 3) return -1;
      ci->filter_id[f] = fid;
      ci->filter_cd0[f] = ncd ? u32(d + pos) : 0;
      pos += 4u * (size_t)ncd;
      if (version == 1 && (ncd & 1)) pos += 4;
    }
    ci->nfilters = nf;
  }
  return 0;
}

/* Parses the object header at `addr` (relative to `base`).  0 = a numeric vector this reader takes. */
static int parse_cell(fd_t fd, uint64_t base, uint64_t addr, cell_info *ci) {
  uint8_t head[16];
  memset(ci, 0, sizeof *ci);
  if (read_at(fd, base + addr, head, 16)) return -1;
  if (head[0] != 1) return -1; /* "OHDR" (version 2) and anything else: the Python reader's business */
  int nmsg = u16(head + 2), got = 0, seen = 0;
  uint64_t block_at[8], block_len[8];
  int nblocks = 1, cur = 0;
  block_at[0] = addr + 16;
  block_len[0] = u32(head + 8);
  while (cur < nblocks && got < nmsg) {
    uint64_t blen = block_len[cur];
    if (blen > (1u << 20)) return -1;
    uint8_t *buf = (uint8_t *)malloc(blen ? blen : 1);
    if (!buf) return -1;
    if (read_at(fd, base + block_at[cur], buf, blen)) {
      free(buf);
      return -1;
    }
    uint64_t pos = 0;
    while (pos + 8 <= blen && got < nmsg) {
      int type = u16(buf + pos), msize = u16(buf + pos + 2);
      if (pos + 8 + (uint64_t)msize > blen) {
        free(buf);
        return -1;
      }
      const uint8_t *d = buf + pos + 8;
      /* message flags, bit 1: the message is SHARED -- its body is a pointer to the real message (a
       * committed datatype, say), not the message.  Not this reader's subset: the Python reader's. */
      if ((buf[pos + 4] & 2) && (type == 0x0001 || type == 0x0003 || type == 0x0008 || type == 0x000B)) {
        free(buf);
        return -1;
      }
      if (type == 0x0010) { /* continuation */
        if (msize < 16 || nblocks == 8) {
          free(buf);
          return -1;
        }
        block_at[nblocks] = u64(d);
        block_len[nblocks] = u64(d + 8);
        ++nblocks;
      } else if (take_message(ci, type, d, (size_t)msize, base + block_at[cur] + pos + 8, &seen)) {
        free(buf);
        return -1;
      }
      pos += 8 + (uint64_t)msize;
      ++got;
    }
    free(buf);
    ++cur;
  }
  if (seen != 7 || ci->elem_size < 1 || ci->elem_size > 16) return -1;
  /* a vector: rank >= 2 (MATLAB has no rank-1 arrays; a rank-1 dataset is an empty cell's
   * dimension list) with at most one extent above 1 */
  if (ci->rank < 2) return -1;
  int long_axes = 0;
  for (int i = 0; i < ci->rank; ++i) long_axes += ci->dims[i] > 1;
  if (long_axes > 1) return -1;
  return 0;
}

static uint64_t cell_count(const cell_info *ci) {
  uint64_t c = 1;
  for (int i = 0; i < ci->rank; ++i) c *= ci->dims[i];
  return c;
}

/* Undoes the filter pipeline on one chunk: raw[rawlen] -> out[outlen].  0 on success. */
static int decode_chunk(const cell_info *ci, uint32_t mask, uint8_t *raw, size_t rawlen, uint8_t *out, size_t outlen) {
  uint8_t *cur = raw, *tmp = NULL;
  size_t curlen = rawlen;
  int rc = 0;
  for (int f = ci->nfilters - 1; f >= 0 && !rc; --f) {
    if (mask & (1u << f)) continue;
    if (ci->filter_id[f] == 3) { /* fletcher32: checksum behind the data */
      if (curlen < 4) rc = -1;
      else curlen -= 4;
    } else if (ci->filter_id[f] == 1) { /* deflate */
      uint8_t *dst = (uint8_t *)malloc(outlen ? outlen : 1);
      uLongf dl = (uLongf)outlen;
      if (!dst || uncompress(dst, &dl, cur, (uLong)curlen) != Z_OK) {
        free(dst);
        rc = -1;
        break;
      }
      free(tmp);
      tmp = cur = dst;
      curlen = (size_t)dl;
    } else { /* shuffle: bytes of the elements stored plane by plane */
      size_t es = ci->filter_cd0[f] ? ci->filter_cd0[f] : (size_t)ci->elem_size;
      size_t ne = es ? curlen / es : 0;
      uint8_t *dst = (uint8_t *)malloc(curlen ? curlen : 1);
      if (!dst || !es) {
        free(dst);
        rc = -1;
        break;
      }
      for (size_t b = 0; b < es; ++b)
        for (size_t e = 0; e < ne; ++e) dst[e * es + b] = cur[b * ne + e];
      memcpy(dst + ne * es, cur + ne * es, curlen - ne * es);
      free(tmp);
      tmp = cur = dst;
    }
  }
  if (!rc) {
    if (curlen < outlen) rc = -1;
    else memcpy(out, cur, outlen);
  }
  free(tmp);
  return rc;
}

/* Walks the chunk B-tree; copies every chunk's part of the (one-axis) vector into out. */
/* expect_level: the level this node must have (its parent's minus one), -1 for the root: a child that
 * does not sit exactly one level below its parent is a damaged (possibly cyclic) tree. */
static int read_chunks(fd_t fd, uint64_t base, const cell_info *ci, uint64_t node, int axis, uint8_t *out,
                       int expect_level) {
  if (node == UNDEF_ADDR) return 0;
  uint8_t head[24];
  if (read_at(fd, base + node, head, 24) || memcmp(head, "TREE", 4) || head[4] != 1) return -1;
  int level = head[5], used = u16(head + 6), nd = ci->rank;
  if (expect_level < 0 ? level > 8 : level != expect_level) return -1;
  size_t ksize = 8 + 8 * (size_t)(nd + 1);
  size_t blen = (size_t)used * (ksize + 8) + ksize;
  uint8_t *body = (uint8_t *)malloc(blen);
  if (!body || read_at(fd, base + node + 24, body, blen)) {
    free(body);
    return -1;
  }
  int rc = 0;
  uint64_t chunk_elems = 1;
  for (int i = 0; i < nd; ++i) chunk_elems *= ci->cdims[i];
  for (int e = 0; e < used && !rc; ++e) {
    const uint8_t *k = body + (size_t)e * (ksize + 8);
    uint32_t csize = u32(k), cmask = u32(k + 4);
    uint64_t child = u64(k + ksize);
    if (level > 0) {
      rc = read_chunks(fd, base, ci, child, axis, out, level - 1);
      continue;
    }
    uint64_t off = 0;
    for (int i = 0; i < nd; ++i) {
      uint64_t o = u64(k + 8 + 8 * i);
      if (i == axis) off = o;
      else if (o) rc = -1; /* a chunk off the vector's axis */
    }
    if (rc) break;
    uint64_t n_axis = ci->dims[axis], c_axis = ci->cdims[axis];
    if (off >= n_axis || chunk_elems != c_axis) { /* chunks must be one-axis too */
      rc = -1;
      break;
    }
    uint64_t take = n_axis - off < c_axis ? n_axis - off : c_axis;
    size_t es = (size_t)ci->elem_size;
    uint8_t *raw = (uint8_t *)malloc(csize ? csize : 1), *full = (uint8_t *)malloc((size_t)chunk_elems * es);
    if (!raw || !full || read_at(fd, base + child, raw, csize) ||
        decode_chunk(ci, cmask, raw, csize, full, (size_t)chunk_elems * es))
      rc = -1;
    else
      memcpy(out + off * es, full, (size_t)take * es);
    free(raw);
    free(full);
  }
  free(body);
  return rc;
}

static int read_cell(fd_t fd, uint64_t base, const cell_info *ci, uint8_t *out) {
  uint64_t n = cell_count(ci);
  size_t bytes = (size_t)n * (size_t)ci->elem_size;
  if (!n) return 0;
  if (ci->layout == 0) return ci->size >= bytes ? read_at(fd, ci->compact_at, out, bytes) : -1;
  if (ci->layout == 1) {
    if (ci->addr == UNDEF_ADDR) {
      memset(out, 0, bytes);
      return 0;
    }
    return ci->size >= bytes ? read_at(fd, base + ci->addr, out, bytes) : -1;
  }
  int axis = 0;
  for (int i = 0; i < ci->rank; ++i)
    if (ci->dims[i] > 1) axis = i;
  for (int i = 0; i < ci->rank; ++i)
    if (i != axis && ci->cdims[i] != 1) return -1;
  if (!ci->cdims[axis]) return -1;
  memset(out, 0, bytes);
  return read_chunks(fd, base, ci, ci->addr, axis, out, -1);
}

/* Element counts and element sizes of n cells (object-header addresses relative to `base`, the
 * size of the user block) of the file mapped at `file`: counts[i] = -1 for a cell this reader does
 * not take. */
int gpdla_h5cells_sizes(const uint8_t *file, uint64_t file_len, uint64_t base, const uint64_t *addrs, int64_t n,
                        int64_t *counts, int32_t *elem_sizes, int nthreads) {
  if (!file || !addrs || !counts || !elem_sizes || n < 0) return -1;
  const h5file hf = {file, file_len};
  fd_t fd = &hf;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int64_t i = 0; i < n; ++i) {
    cell_info ci;
    if (parse_cell(fd, base, addrs[i], &ci)) {
      counts[i] = -1;
      elem_sizes[i] = 0;
    } else {
      counts[i] = (int64_t)cell_count(&ci);
      elem_sizes[i] = ci.elem_size;
    }
  }
  return 0;
}

/* Reads cell i into out + byte_offsets[i]; it must hold counts[i] elements of elem_size bytes (as
 * gpdla_h5cells_sizes reported for the SAME or a sibling cell array) of the class asked for
 * (want_float: IEEE floating point, else fixed point).  Returns the number of cells
 * that could not be read as asked (their bytes are left untouched); status[i] = 0 / -1 per cell. */
int64_t gpdla_h5cells_read(const uint8_t *file, uint64_t file_len, uint64_t base, const uint64_t *addrs, int64_t n,
                           int32_t elem_size, int32_t want_float, void *out, const int64_t *byte_offsets,
                           const int64_t *counts, int8_t *status, int nthreads) {
  if (!file || !addrs || !out || !byte_offsets || !counts || !status || n < 0) return -1;
  const h5file hf = {file, file_len};
  fd_t fd = &hf;
  int64_t failed = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : failed)
#endif
  for (int64_t i = 0; i < n; ++i) {
    cell_info ci;
    int rc = parse_cell(fd, base, addrs[i], &ci);
    /* bytes are copied, never converted: the stored type must BE the array's type (an int64 cell
     * must not land bit for bit in a float64 array -- the Python reader converts such a cell) */
    if (!rc && ((int64_t)cell_count(&ci) != counts[i] || ci.elem_size != elem_size ||
                ci.type_class != (want_float ? 1 : 0)))
      rc = -1;
    if (!rc) rc = read_cell(fd, base, &ci, (uint8_t *)out + byte_offsets[i]);
    status[i] = rc ? -1 : 0;
    failed += rc ? 1 : 0;
  }
  return failed;
}
