// raytracer/imageplane.h -- observer's image plane, traced backwards in time.
//
// API contract of the reference's src/raytracer/imageplane.h:14-61: constructor, init helper, redshift
// conveniences, using-declaration and the four index/coordinate helpers.  Implementation: imageplane.cpp.
#ifndef IMAGEPLANE_H_
#define IMAGEPLANE_H_

#include "raytracer.h"

template <typename T>
class ImagePlane : public Raytracer<T> {
public:
    // Plane at distance `dist`, inclination `inc` (degrees) and azimuth `phi`; one ray per (x, y) grid point of
    // [x0, xmax] x [y0, ymax].  The spin is stored NEGATED (time reversal); redshift*(..., reverse = true) undoes it.
    ImagePlane(T dist,
               T inc,
               T x0,
               T xmax,
               T dx,
               T y0,
               T ymax,
               T dy,
               T spin,
               T phi,
               T precision = PRECISION);

    void init_image_plane(T D, T incl, T phi0, T x0, T xmax, T dx, T y0, T ymax, T dy);

    void redshift_start();               // observer at rest, reverse = true
    void redshift(bool projradius);      // Keplerian disc material, reverse = true
    using Raytracer<T>::redshift;

    // ray array index <-> grid position (row-major in x: ix = i * Ny + j)
    inline int get_x_index(int ix) { return static_cast<int>(ix / Ny); }
    inline int get_y_index(int ix) { return ix % Ny; }
    inline T ray_x(int ix) { return m_x0 + get_x_index(ix) * m_dx; }
    inline T ray_y(int ix) { return m_y0 + get_y_index(ix) * m_dy; }

private:
    int Nx, Ny;
    T m_x0, m_xmax, m_dx;
    T m_y0, m_ymax, m_dy;
    T D;
    T incl;
    T phi0;
};

#endif /* IMAGEPLANE_H_ */
