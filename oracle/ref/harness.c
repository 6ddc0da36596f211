/* oracle/_ref harness registry -- TEST INFRASTRUCTURE ONLY.
 *
 * The generated Fortran file (oracle/ref/gen_harness.py) calls orc_reg_ once
 * per COMMON-block variable of the compiled reference.  We only record the
 * base address and the array descriptor so that oracle/refmodel.py can build
 * numpy views.  Nothing here is reference code.
 *
 * Fortran calling convention (flang, implicit interface): every argument by
 * reference, CHARACTER length appended by value at the end of the list.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define ORC_MAX 4096
#define ORC_NAME 64

typedef struct {
  char name[ORC_NAME];
  void *addr;
  int elem_bytes;
  int typecode; /* 1 real, 2 integer, 3 logical */
  int rank;
  int shape[7];
  int lbound[7];
} orc_entry;

static orc_entry table[ORC_MAX];
static int count = 0;

void orc_reg_(const char *name, void *addr, const int *elem_bytes,
              const int *typecode, const int *rank, const int *shape,
              const int *lb, size_t namelen) {
  if (count >= ORC_MAX) return;
  orc_entry *e = &table[count++];
  size_t n = namelen < ORC_NAME - 1 ? namelen : ORC_NAME - 1;
  memcpy(e->name, name, n);
  e->name[n] = 0;
  e->addr = addr;
  e->elem_bytes = *elem_bytes;
  e->typecode = *typecode;
  e->rank = *rank;
  for (int i = 0; i < 7; ++i) {
    e->shape[i] = i < *rank ? shape[i] : 1;
    e->lbound[i] = i < *rank ? lb[i] : 1;
  }
}

int orc_count(void) { return count; }
void orc_reset(void) { count = 0; }

int orc_get(int i, char *name, void **addr, int *elem_bytes, int *typecode,
            int *rank, int *shape, int *lb) {
  if (i < 0 || i >= count) return -1;
  const orc_entry *e = &table[i];
  strcpy(name, e->name);
  *addr = e->addr;
  *elem_bytes = e->elem_bytes;
  *typecode = e->typecode;
  *rank = e->rank;
  for (int k = 0; k < 7; ++k) {
    shape[k] = e->shape[k];
    lb[k] = e->lbound[k];
  }
  return 0;
}
