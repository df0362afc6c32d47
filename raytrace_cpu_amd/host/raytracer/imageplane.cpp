// raytracer/imageplane.cpp -- ImagePlane<T>; reference src/raytracer/imageplane.cpp:11-140.
// Quirks kept because they decide the numbers the applications see: the x coordinate of column i is
// x0 + i*dy (dy, :43); the spin used for the metric here is a double copy of the (negated) member (:38);
// k, h, Q come from the impact-parameter formulae (:100-113), not from calculate_constants_from_p().
#include "imageplane.h"

template <typename T>
ImagePlane<T>::ImagePlane(T dist, T inc, T x0, T xmax, T dx, T y0, T ymax, T dy, T spin, T phi, T precision)
    : Raytracer<T>((((xmax - x0) / dx) + 1) * (((ymax - y0) / dy) + 1), -1 * spin, precision),
      Nx(((xmax - x0) / dx) + 1), Ny(((ymax - y0) / dy) + 1),
      m_x0(x0), m_xmax(xmax), m_dx(dx), m_y0(y0), m_ymax(ymax), m_dy(dy), D(dist), incl(inc), phi0(phi)
{
    init_image_plane(D, incl * M_PI / 180, phi0, x0, xmax, dx, y0, ymax, dy);
}

template <typename T>
void ImagePlane<T>::init_image_plane(T D, T incl, T phi0, T x0, T xmax, T dx, T y0, T ymax, T dy)
{
    const int nx = ((xmax - x0) / dx) + 1;
    const int ny = ((ymax - y0) / dy) + 1;
    const double a = Raytracer<T>::spin;
    Ray<T>* rays = Raytracer<T>::rays;

    // every pixel is a pure function of (i, j): columns are shared among the host threads
#pragma omp parallel for schedule(static) num_threads(kr_host_threads())
    for (int i = 0; i < nx; i++) {
        const T x = x0 + i * dy;
        for (int j = 0; j < ny; j++) {
            Ray<T>& R = rays[i * ny + j];
            const T y = y0 + j * dy;

            // position of the grid point and a momentum pointing along the line of sight
            const T r = sqrt(D * D + x * x + y * y);
            const T theta = acos((D * cos(incl) + y * sin(incl)) / r);
            const T phi = phi0 + atan2(x, D * sin(incl) - y * cos(incl));
            const T pr = D / r;
            const T ptheta = sin(acos(D / r)) / r;
            const T pphi = x * sin(incl) / (x * x + (D * sin(incl) - y * cos(incl)) * (D * sin(incl) - y * cos(incl)));

            // null condition g_ab p^a p^b = 0 solved for pt (positive root)
            const T rhosq = r * r + (a * cos(theta)) * (a * cos(theta));
            const T delta = r * r - 2 * r + a * a;
            const T sigmasq = (r * r + a * a) * (r * r + a * a) - a * a * delta * sin(theta) * sin(theta);
            const T e2nu = rhosq * delta / sigmasq;
            const T e2psi = sigmasq * sin(theta) * sin(theta) / rhosq;
            const T omega = 2 * a * r / sigmasq;
            const T g00 = e2nu - omega * omega * e2psi;
            const T g03 = omega * e2psi;
            const T g11 = -rhosq / delta;
            const T g22 = -rhosq;
            const T g33 = -e2psi;
            const T A = g00;
            const T B = 2 * g03 * pphi;
            const T C = g11 * pr * pr + g22 * ptheta * ptheta + g33 * pphi * pphi;
            T pt = (-B + sqrt(B * B - 4 * A * C)) / (2 * A);
            if (pt < 0) pt = (-B - sqrt(B * B - 4 * A * C)) / (2 * A);

            R.t = 0;
            R.r = r;
            R.theta = theta;
            R.phi = phi;
            R.pt = pt;
            R.pr = pr;
            R.ptheta = ptheta;
            R.pphi = pphi;

            Raytracer<T>::calculate_constants_from_p(i * ny + j, pt, pr, ptheta, pphi);
            R.rdot_sign = -1;
            R.thetadot_sign = 1;

            // constants of motion from the impact parameters of the pixel
            R.k = 1;
            const T b = sqrt(x * x + y * y);
            T beta = asin(y / b);
            if (x < 0) beta = M_PI - beta;
            const T h = -1. * b * sin(incl) * cos(beta);
            const T ltheta = b * sin(beta);
            const T Q = (ltheta * ltheta) - (a * cos(theta)) * (a * cos(theta)) + ((h / tan(theta))) * ((h / tan(theta)));
            R.h = h;
            R.Q = Q;
            R.thetadot_sign = (ltheta >= 0) ? 1 : -1;

            R.steps = 0;
            R.alpha = x;
            R.beta = y;
        }
    }
}

template <typename T>
void ImagePlane<T>::redshift_start()
{
    Raytracer<T>::redshift_start(0, true);
}

template <typename T>
void ImagePlane<T>::redshift(bool projradius)
{
    Raytracer<T>::redshift(-1, true, projradius);
}

template class ImagePlane<double>;
template class ImagePlane<float>;
