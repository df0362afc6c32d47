/* tests/cpp/abi_c_check.c -- include/kr_trace.h is a C header: it must compile as C99 and link from C.
 * Compiled and run by tests/test_capi_symbols.py (no GPU needed: only entry points that do not touch a device are called). */
#include <stdio.h>
#include <string.h>

#include "../../include/kr_trace.h"

int main(void)
{
    kr_params p;
    kr_pointsource s;
    kr_params_default(&p, 0.998);
    memset(&s, 0, sizeof s);
    s.pos[1] = 10.0; s.pos[2] = 1e-3; s.spin = 0.998; s.tol = 100.0; s.E = 1.0;
    s.dcosalpha = 0.05; s.dbeta = 0.05; s.cosalpha0 = -0.995; s.cosalphamax = 0.995; s.beta0 = -3.141592653589793; s.betamax = 3.141592653589793;
    printf("abi %d sizeof(ray_f64) %zu sizeof(ray_f32) %zu sizeof(params) %zu sizeof(stats) %zu\n", kr_abi_version(), sizeof(kr_ray_f64), sizeof(kr_ray_f32),
           sizeof(kr_params), sizeof(kr_stats));
    printf("horizon %.17g isco %.17g rays %lld precision %g integrator %d\n", kr_kerr_horizon(0.998), kr_kerr_isco(0.998, 1),
           (long long) kr_pointsource_count(&s, NULL, NULL), p.precision, p.integrator);
    return (sizeof(kr_ray_f64) == 144 && sizeof(kr_ray_f32) == 84 && kr_abi_version() == KR_ABI_VERSION) ? 0 : 1;
}
