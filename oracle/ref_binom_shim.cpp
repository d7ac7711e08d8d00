// ref_binom_shim.cpp -- OUR few lines around the reference's binomial table (Planner/src/binomial_coefs.cpp, compiled from where it
// lies under /root/reference by oracle/Makefile into _ref/libbinomial_ref.so): the 13 x 13 table c(n, k) as the reference's
// traj_postprocessing node sees it (traj_postprocessing.cpp:29-90).  Test infrastructure only (tests/golden/make_golden.py).
#include "pointcloudTraj/binomial_coefs.h"

extern "C" void refbinom_table(int *out /* [13][13] */)
{
    binomial_coefs b;
    for (int n = 0; n < 13; n++)
        for (int k = 0; k < 13; k++) out[n * 13 + k] = b.c(n, k);
}
