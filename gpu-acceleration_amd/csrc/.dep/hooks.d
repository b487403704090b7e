../libmsm_hip_hooks.so: msm_hip.hip ../../include/msm_hip.h host_g1.hpp \
  msm_host_pool.hpp msm_planner.hpp msm_kernels.hpp ec_bn254.hpp \
  fp_bn254.hpp fp29_constants.inc ec_wide.hpp glv_bn254.hpp \
  msm_multi.inc msm_testhooks.inc ../../include/msm_hip_testhooks.h \
  msm_testhooks_kernels.hpp
