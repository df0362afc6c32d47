// raytracer/pointsource.cpp -- PointSource<T>; reference src/raytracer/pointsource.cpp:11-83.
// Grid bookkeeping follows the reference exactly because it fixes which array slots are rays:
//   nRays      = int( ((cmax-c0)/dc + 1) * ((bmax-b0)/db + 1) )   -- truncated PRODUCT of doubles (:12)
//   n_cosalpha = int( (cmax-c0)/dc + 1 ),  n_beta likewise         -- truncated factors (:16-17)
// so nRays can exceed n_cosalpha*n_beta; the surplus slots keep steps = -1 and are never traced.
#include "pointsource.h"

template <typename T>
PointSource<T>::PointSource(T* pos, T V, T spin, T tol, T dcosalpha, T dbeta, T cosalpha0, T cosalphamax, T beta0, T betamax, T E)
    : Raytracer<T>((((cosalphamax - cosalpha0) / dcosalpha) + 1) * (((betamax - beta0) / dbeta) + 1), spin, tol), velocity(V), energy(E)
{
    n_cosalpha = ((cosalphamax - cosalpha0) / dcosalpha) + 1;
    n_beta = ((betamax - beta0) / dbeta) + 1;
    init_pointsource(pos, dcosalpha, dbeta, cosalpha0, cosalphamax, beta0, betamax);
}

template <typename T>
void PointSource<T>::init_pointsource(T* pos, T dcosalpha, T dbeta, T cosalpha0, T cosalphamax, T beta0, T betamax)
{
    Ray<T>* rays = Raytracer<T>::rays;
    // every slot is a pure function of (i, j): rows are shared among the host threads (the reference's loop is serial;
    // at 1e7 rays it costs more than the whole GPU trace)
#pragma omp parallel for schedule(static) num_threads(kr_host_threads())
    for (int i = 0; i < n_cosalpha; i++) {
        const T cosalpha = cosalpha0 + i * dcosalpha;
        for (int j = 0; j < n_beta; j++) {
            Ray<T>& R = rays[i * n_beta + j];
            const T beta = beta0 + j * dbeta;
            if (cosalpha >= cosalphamax || beta >= betamax) {   // outside the half-open ranges: slot stays unused
                R.steps = -1;
                continue;
            }
            R.alpha = cosalpha;     // the reference stores cos(alpha) in `alpha` (:48); consumers depend on it
            R.beta = beta;
            R.t = pos[0];
            R.r = pos[1];
            R.theta = pos[2];
            R.phi = pos[3];
            R.pt = R.pr = R.ptheta = R.pphi = 0;
            R.steps = 0;
            Raytracer<T>::calculate_constants(i * n_beta + j, acos(cosalpha), beta, velocity, energy);
        }
    }
}

template <typename T>
void PointSource<T>::redshift_start()
{
    Raytracer<T>::redshift_start(velocity);
}

template <typename T>
void PointSource<T>::redshift(T V)
{
    Raytracer<T>::redshift(V);
}

template class PointSource<double>;
template class PointSource<float>;
