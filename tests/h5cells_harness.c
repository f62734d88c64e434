#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
int gpdla_h5cells_sizes(const uint8_t *, uint64_t, uint64_t, const uint64_t *, int64_t, int64_t *, int32_t *, int);
int64_t gpdla_h5cells_read(const uint8_t *, uint64_t, uint64_t, const uint64_t *, int64_t, int32_t, int32_t, void *, const int64_t *, const int64_t *, int8_t *, int);
int main(int argc, char **argv) {
  int fd = open(argv[1], O_RDONLY); struct stat st; fstat(fd, &st);
  const uint8_t *m = mmap(0, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  FILE *f = fopen(argv[2], "rb"); uint64_t addrs[256]; int64_t n = fread(addrs, 8, 256, f); fclose(f);
  int64_t counts[256]; int32_t es[256];
  gpdla_h5cells_sizes(m, st.st_size, 512, addrs, n, counts, es, 4);
  long ok = 0, bad = 0;
  for (int64_t i = 0; i < n; i++) {
    if (counts[i] < 0) { bad++; continue; }
    void *out = malloc(counts[i] * es[i] + 1); int64_t off = 0; int8_t status;
    int64_t failed = gpdla_h5cells_read(m, st.st_size, 512, addrs + i, 1, es[i], es[i] == 8, out, &off, counts + i, &status, 1);
    ok += !failed; bad += !!failed; free(out);
  }
  /* truncated file: every length from 0 to a few KB and a few around the end */
  for (uint64_t len = 0; len < 6000 && len < (uint64_t)st.st_size; len += 37)
    gpdla_h5cells_sizes(m, len, 512, addrs, n, counts, es, 2);
  for (int i = 0; i < n; i++) if (counts[i] > 0) { /* wrong element size / count must be refused */
    void *out = malloc(counts[i] * 16 + 16); int64_t off = 0, c2 = counts[i] + 1; int8_t status;
    gpdla_h5cells_read(m, st.st_size, 512, addrs + i, 1, es[i], es[i] == 8, out, &off, &c2, &status, 1);
    if (status != -1) { printf("count mismatch accepted!\n"); return 1; }
    /* the stored class must be the class asked for: a float cell is not an integer array's bytes */
    gpdla_h5cells_read(m, st.st_size, 512, addrs + i, 1, es[i], es[i] != 8, out, &off, counts + i, &status, 1);
    if (status != -1) { printf("type class mismatch accepted!\n"); return 1; }
    free(out); break; }
  printf("%s: ok %ld, refused %ld of %ld\n", argv[1], ok, bad, (long)n);
  return 0;
}
