static int predict_moments_chunk(gpmpc_gp* h, int method, int B, const double* dZ, const double* dSigma,
                                 double* dMean, double* dCov) {
    (void)h; (void)method; (void)B; (void)dZ; (void)dSigma; (void)dMean; (void)dCov;
    return fail(GPMPC_EINVAL, "moment-based methods are not built yet");
}
