// raytracer/imageplane.cpp -- ImagePlane<T>; reference src/raytracer/imageplane.cpp:11-140.
// Quirks kept because they decide the numbers the applications see: the x coordinate of column i is
// x0 + i*dy (dy, :43); the spin used for the metric here is a double copy of the (negated) member (:38);
// k, h, Q come from the impact-parameter formulae (:100-113), not from calculate_constants_from_p().
#include "imageplane.h"

#include "image_ray.h"

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
    const double a = Raytracer<T>::spin;          // (a double copy, whatever T is: :38)
    Ray<T>* rays = Raytracer<T>::rays;

    // every pixel is a pure function of (i, j): columns are shared among the host threads
#pragma omp parallel for schedule(static) num_threads(kr_host_threads())
    for (int i = 0; i < nx; i++) {
        const T x = x0 + i * dy;                  // (dy: :43)
        for (int j = 0; j < ny; j++) {
            const int ix = i * ny + j;
            const T y = y0 + j * dy;
            const krhost::CameraRay<T> c = krhost::camera_ray<T, double, true, false>(D, incl, phi0, x, y, a);
            Ray<T>& R = rays[ix];
            R.t = 0;
            R.r = c.r; R.theta = c.theta; R.phi = c.phi;
            R.pt = c.pt; R.pr = c.pr; R.ptheta = c.ptheta; R.pphi = c.pphi;
            // the signs and flip counter as calculate_constants_from_p() leaves them, its k, h, Q replaced by the impact-parameter values (:96-113)
            Raytracer<T>::calculate_constants_from_p(ix, c.pt, c.pr, c.ptheta, c.pphi);
            R.rdot_sign = -1;
            R.k = 1;
            R.h = c.h;
            R.Q = c.Q;
            R.thetadot_sign = c.thetadot_sign;
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
