// apps/emissivity_table.h -- the 7-column table of the reference's emissivity programs (emissivity.cpp:128-147):
// r, area, count, flux/area, emis/area, <g> = sum_g/count, <t> = sum_t/count per radial bin, TextOutput format
// (width 20, scientific, 8 digits; the count column as an integer; empty bins give nan from 0/0).
#ifndef KR_EMISSIVITY_TABLE_H_
#define KR_EMISSIVITY_TABLE_H_

#include <string>

#include "../host/include/text_output.h"

namespace krapp {

// raw accumulators in, table out
inline void write_emissivity_table(const std::string& out_name, int Nr, const double* bin_r, const double* bin_area, const double* count,
                                   const double* sum_flux, const double* sum_emis, const double* sum_g, const double* sum_t)
{
    TextOutput outfile(out_name.c_str());
    for (int ir = 0; ir < Nr; ++ir) {
        const long rays = static_cast<long>(count[ir]);
        const double flux = sum_flux[ir] / bin_area[ir];
        const double emis = sum_emis[ir] / bin_area[ir];
        const double mean_g = sum_g[ir] / rays;      // 0/0 -> NaN in empty bins, as in the reference's table
        const double mean_t = sum_t[ir] / rays;
        outfile << bin_r[ir] << bin_area[ir] << rays << flux << emis << mean_g << mean_t << endl;
    }
    outfile.close();
}

}   // namespace krapp

#endif /* KR_EMISSIVITY_TABLE_H_ */
