// The compiled kernel set.  predict_kernel<T, D, NB> is compiled for every D in
// GP_FOR_EACH_KERNEL_D and every NB in GP_FOR_EACH_KERNEL_NB (NB = 16-row blocks of
// training points, so N_train <= 16*NB); a caller's (N, D) runs on the smallest kernel
// that holds it, the unused dimensions / training rows being zero padding that contributes
// exactly 0 to every sum.  build.py compiles one translation unit per (dtype, NB).
#pragma once
#define GP_FOR_EACH_KERNEL_D(X) X(2) X(4) X(5) X(8) X(10) X(11) X(12) X(16)
#define GP_FOR_EACH_KERNEL_NB(X) X(2) X(4) X(7) X(8) X(12) X(16) X(19) X(20)
#define GP_MAX_KERNEL_D 16
#define GP_MAX_KERNEL_NB 20
