/* ref_shim.c — thin ctypes-friendly wrappers, appended (on stdin, never written to
 * disk) to the C text that oracle/Makefile streams out of
 * /root/reference/doc/filters004.txt (lines 1-115: FILTER struct + iir_filter();
 * lines 217-413: prewarp()/bilinear()/szxform()).  Only this shim is ours; the
 * reference text is compiled from where it lies and only the .so lands in oracle/_ref/.
 * TEST INFRASTRUCTURE ONLY (see oracle_dsp.hpp).
 *
 * iir_filter() is a K&R definition, so its float argument travels as double; the
 * wrapper hides that. */
void ref_szxform(double a0, double a1, double a2, double b0, double b1, double b2,
                 double fc, double fs, double *k, float *coef4) {
  szxform(&a0, &a1, &a2, &b0, &b1, &b2, fc, fs, k, coef4);
}

/* coef = [gain, (beta1, beta2, alpha1, alpha2) x sections]; runs n samples. */
void ref_iir_run(float *coef, unsigned sections, const float *x, float *y, unsigned n) {
  FILTER f;
  unsigned i;
  f.length = sections;
  f.history = 0;
  f.coef = coef;
  for (i = 0; i < n; i++) y[i] = iir_filter(x[i], &f);
  free(f.history);
}
