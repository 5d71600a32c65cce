/* alloc_inject.h — TEST ONLY, force-included (gcc -include) in front of the product's host sources:
 * routes their allocations through a counting allocator that can be told to fail the k-th one
 * (tests/host_sanitize/alloc_inject.c). */
#ifndef STORM_ALLOC_INJECT_H_
#define STORM_ALLOC_INJECT_H_
#include <stddef.h>
#include <stdlib.h>
void* inject_malloc(size_t n);
void* inject_calloc(size_t a, size_t b);
void* inject_realloc(void* p, size_t n);
int inject_posix_memalign(void** out, size_t align, size_t n);
#define malloc(n) inject_malloc(n)
#define calloc(a, b) inject_calloc(a, b)
#define realloc(p, n) inject_realloc(p, n)
#define posix_memalign(o, a, n) inject_posix_memalign(o, a, n)
#endif
