// raytracer/pointsource.h -- isotropic point source orbiting the rotation axis; API of the reference's
// src/raytracer/pointsource.h:14-33 (same constructor signature and defaults, same member functions).
#ifndef POINTSOURCE_H_
#define POINTSOURCE_H_

#include "raytracer.h"

template <typename T>
class PointSource : public Raytracer<T> {
private:
    T energy;
    T velocity;
    int n_cosalpha;
    int n_beta;

public:
    // rays leave the source at polar angle alpha (from the local radial direction) and azimuth beta, on a regular
    // grid in (cos alpha, beta); `tol` is the Raytracer step precision
    PointSource(T* pos, T V, T spin, T tol, T dcosalpha, T dbeta, T cosalpha0 = -0.999999, T cosalphamax = 0.995,
                T beta0 = -0.995 * M_PI, T betamax = M_PI, T E = 1);

    void init_pointsource(T* pos, T dcosalpha, T dbeta, T cosalpha0 = -0.999999, T cosalphamax = 0.995, T beta0 = -0.995 * M_PI,
                          T betamax = M_PI);

    void redshift_start();   // emitted energies in the frame of the source (angular velocity V)
    void redshift(T V);
    using Raytracer<T>::redshift;
};

#endif /* POINTSOURCE_H_ */
