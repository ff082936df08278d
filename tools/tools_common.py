"""Names shared by the developer tools."""
PHASE_NAMES = ["setup", "vis_eval", "vis_gather", "lm", "schur", "zero", "imu_raw", "imu_whiten", "imu_gather", "prior", "cost_red",
               "fin_scale", "fin_cauchy", "fin_pass", "chol_diag", "chol_trsm", "chol_upd", "back", "lm_back", "dogleg", "plus", "norms", "other",
               "chain_fwd", "chain_bwd", "ch_T(wave3)", "ch_owner(w0)", "ch_mfma(w2)", "ch_interval"]
