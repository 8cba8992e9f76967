// hosek.h — the RGB variant of the Hosek-Wilkie analytic sky-dome radiance model, as the reference's "sky" light uses it.
//
// L. Hosek, A. Wilkie, "An Analytic Model for Full Spectral Sky-Dome Radiance", ACM TOG 31(4), 2012. The reference ships the
// authors' sample implementation 1.4a (src/runtime/skysun/model/ArHosekSkyModel.cpp) and calls two entry points of it from
// SkyModel::SkyModel (src/runtime/skysun/SkyModel.cpp:9-54): arhosek_rgb_skymodelstate_alloc_init(turbidity, albedo, elevation)
// and arhosek_tristim_skymodel_radiance(state, theta, gamma, channel). Both are restated here from the paper's formulas as that
// implementation evaluates them (double precision, libm):
//   * the nine coefficients of the radiance distribution and the mean radiance are quintic Bezier curves over the cube root of the
//     normalised solar elevation, fitted per turbidity (1 .. 10) and ground albedo (0, 1) and interpolated bilinearly between the
//     four neighbouring fits (ArHosekSkyModel.cpp:147-234);
//   * F(theta, gamma) = (1 + A e^(B / (cos theta + 0.01))) (C + D e^(E gamma) + F cos^2 gamma + G chi(H, gamma) + I sqrt(cos theta)),
//     chi(H, gamma) = (1 + cos^2 gamma) / (1 + H^2 - 2 H cos gamma)^1.5 (:236-250).
// The fitted control points are published data: ignis_amd/data/hosek_rgb.f64 (tools/make_hosek_tables.py), compiled into the
// library as hosek_rgb_tables.inc by the Makefile.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace igh {
namespace hosek {

// per channel: 1080 configuration values (albedo 0 | 1) x (turbidity 1 .. 10) x (6 control points) x (9 coefficients),
// then 120 mean-radiance values (albedo) x (turbidity) x (6 control points)
inline const double* tables()
{
    static const uint64_t bits[3 * 1200] = {
#include "hosek_rgb_tables.inc"
    };
    static double values[3 * 1200];
    static const bool once = [] {
        std::memcpy(values, bits, sizeof(values));
        return true;
    }();
    (void)once;
    return values;
}

// quintic Bernstein basis at x
inline void bernstein5(double x, double w[6])
{
    const double y = 1.0 - x;
    w[0] = std::pow(y, 5.0);
    w[1] = 5.0 * std::pow(y, 4.0) * x;
    w[2] = 10.0 * std::pow(y, 3.0) * std::pow(x, 2.0);
    w[3] = 10.0 * std::pow(y, 2.0) * std::pow(x, 3.0);
    w[4] = 5.0 * y * std::pow(x, 4.0);
    w[5] = std::pow(x, 5.0);
}

struct ChannelState {
    double config[9];
    double radiance;
};

// arhosek_rgb_skymodelstate_alloc_init for ONE channel (SkyModel.cpp builds one state per channel from that channel's ground
// albedo and reads only that channel of it). `solar_elevation` is passed through as the reference passes it.
inline ChannelState init(int channel, double turbidity, double albedo, double solar_elevation)
{
    const double* cfg = tables() + (size_t)channel * 1200;
    const double* rad = cfg + 1080;
    const int it      = (int)turbidity;
    const double rem  = turbidity - (double)it;
    double w[6];
    bernstein5(std::pow(solar_elevation / (3.14159265358979323846 / 2.0), 1.0 / 3.0), w);

    ChannelState s{};
    // the four fits around (albedo, turbidity), in the sample implementation's order: (alb 0, low), (alb 1, low), (alb 0, high), (alb 1, high)
    const struct {
        int alb, turb;
        double weight;
    } corner[4] = { { 0, it - 1, (1.0 - albedo) * (1.0 - rem) }, { 1, it - 1, albedo * (1.0 - rem) }, { 0, it, (1.0 - albedo) * rem }, { 1, it, albedo * rem } };
    for (int c = 0; c < 4; ++c) {
        if (c == 2 && it == 10)
            break; // no fit above turbidity 10
        const double* m = cfg + 9 * 6 * 10 * corner[c].alb + 9 * 6 * corner[c].turb;
        for (int i = 0; i < 9; ++i) {
            const double curve = w[0] * m[i] + w[1] * m[i + 9] + w[2] * m[i + 18] + w[3] * m[i + 27] + w[4] * m[i + 36] + w[5] * m[i + 45];
            s.config[i]        = c == 0 ? corner[c].weight * curve : s.config[i] + corner[c].weight * curve;
        }
        const double* r    = rad + 6 * 10 * corner[c].alb + 6 * corner[c].turb;
        const double curve = w[0] * r[0] + w[1] * r[1] + w[2] * r[2] + w[3] * r[3] + w[4] * r[4] + w[5] * r[5];
        s.radiance         = c == 0 ? corner[c].weight * curve : s.radiance + corner[c].weight * curve;
    }
    return s;
}

// arhosek_tristim_skymodel_radiance: theta = angle from the zenith, gamma = angle to the sun
inline double radiance(const ChannelState& s, double theta, double gamma)
{
    const double* c   = s.config;
    const double expM = std::exp(c[4] * gamma);
    const double rayM = std::cos(gamma) * std::cos(gamma);
    const double mieM = (1.0 + std::cos(gamma) * std::cos(gamma)) / std::pow(1.0 + c[8] * c[8] - 2.0 * c[8] * std::cos(gamma), 1.5);
    const double zen  = std::sqrt(std::cos(theta));
    return (1.0 + c[0] * std::exp(c[1] / (std::cos(theta) + 0.01))) * (c[2] + c[3] * expM + c[5] * rayM + c[6] * mieM + c[7] * zen) * s.radiance;
}

} // namespace hosek
} // namespace igh
