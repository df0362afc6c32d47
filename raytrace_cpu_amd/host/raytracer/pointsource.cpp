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
    // Slot ix = i * n_beta + j of the array is the ray in direction (cos(alpha), beta) = (cosalpha0 + i dcosalpha, beta0 + j dbeta) when that point
    // lies inside the half-open ranges, and an unused slot (steps = -1, as the base constructor left it) otherwise.  Every slot is a pure
    // function of its index, so the slots -- not the rows -- are shared among the host threads (the reference's double loop is serial; at 1e7 rays
    // it costs more than the whole GPU trace).
    const long n_slots = static_cast<long>(n_cosalpha) * n_beta;
#pragma omp parallel for schedule(static) num_threads(kr_host_threads())
    for (long ix = 0; ix < n_slots; ix++) {
        const int i = static_cast<int>(ix / n_beta), j = static_cast<int>(ix % n_beta);
        const T cosalpha = cosalpha0 + i * dcosalpha;
        const T beta = beta0 + j * dbeta;
        Ray<T>& R = rays[ix];
        if (cosalpha >= cosalphamax || beta >= betamax) {
            R.steps = -1;
            continue;
        }
        R.t = pos[0]; R.r = pos[1]; R.theta = pos[2]; R.phi = pos[3];      // every ray starts at the source's event
        R.pt = R.pr = R.ptheta = R.pphi = 0;
        R.alpha = cosalpha;     // sic: the reference keeps cos(alpha) in `alpha` (:48); consumers depend on it
        R.beta = beta;
        R.steps = 0;
        Raytracer<T>::calculate_constants(static_cast<int>(ix), acos(cosalpha), beta, velocity, energy);
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
