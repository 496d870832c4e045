// The compiled kernel set.  predict_kernel<T, D, NK> is compiled for every D in
// GP_FOR_EACH_KERNEL_D and every NK in GP_FOR_EACH_KERNEL_NK (NK = k-steps = groups of 4
// training points, so N_train <= 4*NK; 63 and 75 are the BASELINE shapes N_train = 250 and 300
// exactly); a caller's (N, D) runs on the smallest kernel that holds it, the unused dimensions /
// training rows being zero padding that contributes exactly 0 to every sum.  The matrix-core
// Hessian kernel is compiled per 16-block count NB = ceil(NK / 4).  build.py compiles one
// translation unit per (dtype, NK) and per (dtype, NB).
#pragma once
#define GP_FOR_EACH_KERNEL_D(X) X(2) X(4) X(5) X(8) X(10) X(11) X(12) X(16)
#define GP_FOR_EACH_KERNEL_NK(X) X(8) X(16) X(28) X(32) X(48) X(63) X(64) X(75) X(76) X(80)
#define GP_FOR_EACH_KERNEL_NB(X) X(2) X(4) X(7) X(8) X(12) X(16) X(19) X(20)
#define GP_MAX_KERNEL_D 16
#define GP_MAX_KERNEL_NB 20
