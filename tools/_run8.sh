python tools/ab_kernel.py --rounds 4 build_ab/libtr_r01.so build_ab/libtr_head.so build_ab/libtr_h_est.so build_ab/libtr_h_nosun.so build_ab/libtr_h_both.so 2>&1 | tail -5
