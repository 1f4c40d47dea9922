/* gpcb_io.h -- tiny named-array container ("GPCB1") shared by the oracle drivers and the tests.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gpc_amd/ may include this file.
 *
 * Layout: magic "GPCB1\n", then records
 *     int32 name_len | name bytes | int64 rows | int64 cols | double data[rows*cols]  (column-major)
 * The Python twin is oracle/gpcb.py.
 */
#ifndef GPCB_IO_H
#define GPCB_IO_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define GPCB_MAX_ARRAYS 64

typedef struct {
  char name[64];
  int64_t rows, cols;
  double* data;
} gpcb_array;

typedef struct {
  int n;
  gpcb_array a[GPCB_MAX_ARRAYS];
} gpcb_file;

static inline int gpcb_read(const char* path, gpcb_file* f)
{
  FILE* fp = fopen(path, "rb");
  char magic[6];
  f->n = 0;
  if(!fp) return -1;
  if(fread(magic, 1, 6, fp) != 6 || memcmp(magic, "GPCB1\n", 6) != 0) { fclose(fp); return -2; }
  for(;;) {
    int32_t nl;
    gpcb_array* a;
    if(fread(&nl, sizeof nl, 1, fp) != 1) break;
    if(nl <= 0 || nl >= 64 || f->n >= GPCB_MAX_ARRAYS) { fclose(fp); return -3; }
    a = &f->a[f->n];
    if(fread(a->name, 1, (size_t)nl, fp) != (size_t)nl) { fclose(fp); return -4; }
    a->name[nl] = 0;
    if(fread(&a->rows, 8, 1, fp) != 1 || fread(&a->cols, 8, 1, fp) != 1) { fclose(fp); return -5; }
    a->data = (double*)malloc(sizeof(double) * (size_t)(a->rows * a->cols) + 8);
    if(a->rows * a->cols > 0 &&
       fread(a->data, sizeof(double), (size_t)(a->rows * a->cols), fp) != (size_t)(a->rows * a->cols)) {
      fclose(fp); return -6;
    }
    f->n++;
  }
  fclose(fp);
  return 0;
}

static inline const gpcb_array* gpcb_find(const gpcb_file* f, const char* name)
{
  int i;
  for(i = 0; i < f->n; i++)
    if(strcmp(f->a[i].name, name) == 0) return &f->a[i];
  return NULL;
}

static inline const gpcb_array* gpcb_need(const gpcb_file* f, const char* name)
{
  const gpcb_array* a = gpcb_find(f, name);
  if(!a) { fprintf(stderr, "gpcb: missing array '%s'\n", name); exit(2); }
  return a;
}

static inline FILE* gpcb_open_write(const char* path)
{
  FILE* fp = fopen(path, "wb");
  if(fp) fwrite("GPCB1\n", 1, 6, fp);
  return fp;
}

static inline void gpcb_write(FILE* fp, const char* name, int64_t rows, int64_t cols, const double* data)
{
  int32_t nl = (int32_t)strlen(name);
  fwrite(&nl, sizeof nl, 1, fp);
  fwrite(name, 1, (size_t)nl, fp);
  fwrite(&rows, 8, 1, fp);
  fwrite(&cols, 8, 1, fp);
  if(rows * cols > 0) fwrite(data, sizeof(double), (size_t)(rows * cols), fp);
}

static inline void gpcb_write_scalar(FILE* fp, const char* name, double v)
{
  gpcb_write(fp, name, 1, 1, &v);
}
#endif
