// include/gramschmidt_basis.h -- orthonormal frame of an arbitrary time-like observer by Gram-Schmidt in the Kerr metric
// (API of the reference's src/include/gramschmidt_basis.h:11-123: `vectors[0]` = e_t, then the r-, theta- and
// phi-like legs, right-handed).
//
// Deliberate difference: the reference seeds the three spatial guesses by setting one component each (:41-47) and
// leaves the other nine indeterminate; here they are the coordinate unit vectors (zeros elsewhere), which is what the
// comment there describes.
#ifndef CUDAKERR_GRAMSCHMIDT_BASIS_H
#define CUDAKERR_GRAMSCHMIDT_BASIS_H

#include <cmath>

#include "kerr.h"

template <typename T>
class BasisVectors {
public:
    T vectors[4][4];
};

template <typename T>
class GramSchmidt_Basis : public BasisVectors<T> {
public:
    GramSchmidt_Basis(T* pos, T* vel, T spin)
    {
        kerr_metric(g_, pos, spin);
        orthogonalise(this->vectors, pos, vel);
        normalise();
    }

    void orthogonalise(T (*out)[4], T* /*pos*/, T* vel)
    {
        // seeds: the 4-velocity, then d/dr, d/dtheta, d/dphi (radial first, so that one leg stays close to it)
        double seed[4][4] = {{0, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
        double e[4][4];
        for (int c = 0; c < 4; ++c) seed[0][c] = vel[c];
        for (int i = 0; i < 4; ++i) {
            for (int c = 0; c < 4; ++c) e[i][c] = seed[i][c];
            for (int j = 0; j < i; ++j) {
                // remove the projection on leg j
                const double scale = inner(seed[i], e[j]) / inner(e[j], e[j]);
                for (int c = 0; c < 4; ++c) e[i][c] -= scale * e[j][c];
            }
        }
        // orientation, then reorder (t, phi-like, theta-like, r-like)
        flip_if(e[1], e[1][1] < 0);
        flip_if(e[2], e[2][2] > 0);
        flip_if(e[3], e[3][3] < 0);
        const int order[4] = {0, 3, 2, 1};
        for (int i = 0; i < 4; ++i)
            for (int c = 0; c < 4; ++c) out[i][c] = static_cast<T>(e[order[i]][c]);
    }

    void normalise()
    {
        for (int i = 0; i < 4; ++i) {
            double row[4];
            for (int c = 0; c < 4; ++c) row[c] = this->vectors[i][c];
            const double len = std::sqrt(std::fabs(inner(row, row)));
            for (int c = 0; c < 4; ++c) this->vectors[i][c] /= len;
        }
    }

private:
    double inner(const double* u, const double* v) const
    {
        double s = 0;
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) s += g_[a][b] * u[a] * v[b];
        return s;
    }
    static void flip_if(double* v, bool cond)
    {
        if (cond)
            for (int c = 0; c < 4; ++c) v[c] = -v[c];
    }

    T g_[4][4];
};

#endif /* CUDAKERR_GRAMSCHMIDT_BASIS_H */
