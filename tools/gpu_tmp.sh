tools/gpu_ab_verify.sh r3p score2 medlds triinl score2 medlds triinl
