// raytracer/pointsource.h -- isotropic point source orbiting the rotation axis.
//
// API contract of the reference's src/raytracer/pointsource.h:14-33: constructor signature and defaults, the
// init helper, the two redshift conveniences and the using-declaration.  Implementation: pointsource.cpp (host,
// O(N)); the rays are then integrated on the GPU by the base class.
#ifndef POINTSOURCE_H_
#define POINTSOURCE_H_

#include "raytracer.h"

template <typename T>
class PointSource : public Raytracer<T> {
public:
    // Rays leave `pos` = (t, r, theta, phi) at polar angle alpha from the local radial direction and azimuth beta,
    // on a regular grid in (cos alpha, beta) with half-open ranges; V is the source's angular velocity dphi/dt,
    // `tol` the step precision handed to Raytracer, E the photon energy in the source frame.
    PointSource(T* pos,
                T V,
                T spin,
                T tol,
                T dcosalpha,
                T dbeta,
                T cosalpha0 = -0.999999,
                T cosalphamax = 0.995,
                T beta0 = -0.995 * M_PI,
                T betamax = M_PI,
                T E = 1);

    void init_pointsource(T* pos,
                          T dcosalpha,
                          T dbeta,
                          T cosalpha0 = -0.999999,
                          T cosalphamax = 0.995,
                          T beta0 = -0.995 * M_PI,
                          T betamax = M_PI);

    // emitted energies in the frame of the source (uses the constructor's V)
    void redshift_start();
    // received/emitted ratio for material orbiting at V (-1: Keplerian at the ray's end point)
    void redshift(T V);
    using Raytracer<T>::redshift;

private:
    int n_beta;        // grid points along beta
    int n_cosalpha;    // grid points along cos(alpha)
    T velocity;
    T energy;
};

#endif /* POINTSOURCE_H_ */
