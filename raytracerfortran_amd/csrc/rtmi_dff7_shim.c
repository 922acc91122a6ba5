/*
 * librtmi_dff7.so -- exports `dff_` with the SEVEN-argument signature of the
 * reference's verbose build (/root/reference/subroutineR.f90:405: no keep_delta),
 * the flavour README.md:16-21 and adaptMCMC.Rmd:74-78 call after
 * dyn.load("./subroutineR.so").  It never touches an eighth pointer.
 *
 * include/rtmi.h is deliberately not included: it declares the eight-argument
 * prototype of the same symbol name.
 */
void rtmi_dff7(const double *vels, const double *depths, const int *NLayers, double *src_offset,
               double *src_depth, const int *NSrc, double *timeP); /* include/rtmi.h */

void dff_(const double *vels, const double *depths, const int *NLayers, double *src_offset, double *src_depth,
          const int *NSrc, double *timeP)
{
    rtmi_dff7(vels, depths, NLayers, src_offset, src_depth, NSrc, timeP);
}
