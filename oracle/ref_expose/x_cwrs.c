/* ref_expose/x_cwrs.c — TEST INFRASTRUCTURE: reach celt/cwrs.c statics (icwrs, U table). */
#include "cwrs.c"
const opus_uint32 *ref_pvq_u_data(int *n) { *n = (int)(sizeof(CELT_PVQ_U_DATA) / sizeof(CELT_PVQ_U_DATA[0])); return CELT_PVQ_U_DATA; }
opus_uint32 ref_pvq_u(int n, int k) { return CELT_PVQ_U(n, k); }
opus_uint32 ref_pvq_v(int n, int k) { return CELT_PVQ_V(n, k); }
opus_uint32 ref_icwrs(int n, const int *y) { return icwrs(n, y); }
