/* A process WITHOUT torch: the collective library is resolved by libhpngs itself (dlopen of librccl.so.1), a communicator of
 * one rank is made through each of the two entry points the hosts use (hpn_comm_init: one process per GPU, bench.py;
 * hpn_comm_init_all: the C tools' lanes) and a sum all-reduce over it leaves the vector as it was.  Prints the path of the
 * library that carried it.  (No reference counterpart: the reference is single-process; nearest seam reduceStats,
 * fastq_count_kthread.c:180-210.)  C99, -pedantic -Werror: tests/test_abi_c.py. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "hpngs.h"

static int fail(hpn_ctx *c, const char *what, int rc)
{
    fprintf(stderr, "%s: %s (%s)\n", what, hpn_strerror(rc), c ? hpn_ctx_last_error(c) : "-");
    return 1;
}

int main(void)
{
    hpn_ctx *a = NULL, *b = NULL;
    uint8_t id[HPN_UNIQUE_ID_BYTES];
    uint64_t h[8] = {1, 2, 3, 4, 5, 6, 7, 1ull << 40}, back[8];
    void *d = NULL;
    uint64_t *dv;
    int rc, n = 0;
    if ((rc = hpn_ctx_create(0, &a)) != HPN_OK) return fail(NULL, "hpn_ctx_create", rc);
    if ((rc = hpn_ctx_create(0, &b)) != HPN_OK) return fail(NULL, "hpn_ctx_create", rc);
    if ((rc = hpn_comm_unique_id(id)) != HPN_OK) return fail(a, "hpn_comm_unique_id", rc);
    if ((rc = hpn_comm_init(a, 0, 1, id)) != HPN_OK) return fail(a, "hpn_comm_init", rc);
    if ((rc = hpn_comm_count(a, &n)) != HPN_OK || n != 1) return fail(a, "hpn_comm_count", rc);
    if ((rc = hpn_dev_malloc(a, sizeof h, &d)) != HPN_OK) return fail(a, "hpn_dev_malloc", rc);
    if ((rc = hpn_memcpy_h2d(a, d, h, sizeof h)) != HPN_OK) return fail(a, "hpn_memcpy_h2d", rc);
    if ((rc = hpn_allreduce_u64(a, (uint64_t *)d, 8)) != HPN_OK) return fail(a, "hpn_allreduce_u64", rc);
    if ((rc = hpn_memcpy_d2h(a, back, d, sizeof h)) != HPN_OK || (rc = hpn_ctx_sync(a)) != HPN_OK) return fail(a, "hpn_memcpy_d2h", rc);
    if (memcmp(h, back, sizeof h) != 0) return fail(a, "sum over one rank changed the vector", HPN_E_STATE);
    /* the tools' form: a group of one context */
    if ((rc = hpn_comm_init_all(&b, 1)) != HPN_OK) return fail(b, "hpn_comm_init_all", rc);
    dv = (uint64_t *)d;
    if ((rc = hpn_allreduce_u64_all(&b, &dv, 1, 8)) != HPN_OK) return fail(b, "hpn_allreduce_u64_all", rc);
    if ((rc = hpn_memcpy_d2h(a, back, d, sizeof h)) != HPN_OK || (rc = hpn_ctx_sync(a)) != HPN_OK) return fail(a, "hpn_memcpy_d2h", rc);
    if (memcmp(h, back, sizeof h) != 0) return fail(b, "grouped sum over one rank changed the vector", HPN_E_STATE);
    printf("%s\n", hpn_comm_library());
    hpn_comm_destroy(a);
    hpn_comm_destroy(b);
    hpn_dev_free(a, d);
    hpn_ctx_destroy(b);
    hpn_ctx_destroy(a);
    return 0;
}
