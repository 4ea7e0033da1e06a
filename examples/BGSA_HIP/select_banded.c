/* Linked into the banded build of the drop-in example: the one line a banded/BGSA_HIP main() adds
 * (INTEGRATION.md §2), done here as a constructor so that the reference's main.c stays untouched. */
#include <stdio.h>
#include <stdlib.h>

int bgsa_hip_select_algorithm(int algo);
const char *bgsa_hip_last_error(void);

__attribute__((constructor)) static void select_banded(void)
{
    if (bgsa_hip_select_algorithm(1 /* BGSA_ALGO_BANDED */) != 0) {
        fprintf(stderr, "%s\n", bgsa_hip_last_error());
        exit(1);
    }
}
