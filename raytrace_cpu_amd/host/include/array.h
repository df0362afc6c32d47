// include/array.h -- contiguous 1-, 2- and 3-dimensional arrays with the interface the reference's applications use
// (src/include/array.h: Array :16, Array2D :112, Array3D :226): public `ptr` / extents, row pointers so that `a[i][j]`
// and `T**` conversions work, zero(), raw write/read, and element-wise `/=` and `+=` by a scalar or by another array
// of any element type (0/0 -> NaN is relied upon by imageplane_disc_image.cpp:170-174).
//
// Storage is one block per array; the 2-D and 3-D classes add a table of row pointers into it.
#ifndef ARRAY_H_
#define ARRAY_H_

#include <fstream>
#include <iostream>

namespace krhost {
// shared element-wise kernels over a flat block
template <typename A, typename B>
inline void divide_each(A* a, const B* b, long n)
{
    for (long i = 0; i < n; ++i) a[i] /= b[i];
}
template <typename A, typename B>
inline void add_each(A* a, const B* b, long n)
{
    for (long i = 0; i < n; ++i) a[i] += b[i];
}
template <typename A, typename S>
inline void divide_all(A* a, S s, long n)
{
    for (long i = 0; i < n; ++i) a[i] /= s;
}
template <typename A>
inline void clear_all(A* a, long n)
{
    for (long i = 0; i < n; ++i) a[i] = 0;
}
}   // namespace krhost

#define KR_ARRAY_SCALAR_DIVISIONS(block, count)                                  \
    void operator/=(float s) { krhost::divide_all(block, s, count); }           \
    void operator/=(double s) { krhost::divide_all(block, s, count); }          \
    void operator/=(int s) { krhost::divide_all(block, s, count); }             \
    void operator/=(long s) { krhost::divide_all(block, s, count); }

template <typename T>
class Array {
public:
    T* ptr;
    int num;

    explicit Array(int N, bool init_zero = true) : ptr(new T[N]), num(N)
    {
        if (init_zero) zero();
    }
    ~Array() { delete[] ptr; }
    Array(const Array&) = delete;
    Array& operator=(const Array&) = delete;

    T& operator[](int i) { return ptr[i]; }
    operator T*() { return ptr; }
    int len() { return num; }
    void zero() { krhost::clear_all(ptr, num); }

    void write(std::ofstream* out) { out->write(reinterpret_cast<char*>(ptr), sizeof(T) * num); }
    void read(std::ifstream* in) { in->read(reinterpret_cast<char*>(ptr), sizeof(T) * num); }

    template <typename U>
    void operator/=(Array<U>& o)
    {
        if (o.num != num) { std::cerr << "Array ERROR: Cannot divide arrays with different dimensions"; return; }
        krhost::divide_each(ptr, o.ptr, num);
    }
    template <typename U>
    void operator+=(Array<U>& o)
    {
        if (o.num != num) { std::cerr << "Array ERROR: Cannot add arrays with different dimensions"; return; }
        krhost::add_each(ptr, o.ptr, num);
    }
    KR_ARRAY_SCALAR_DIVISIONS(ptr, num)
};

template <typename T>
class Array2D {
public:
    T** ptr;            // ptr[ix] -> row ix of num_y elements; ptr[0] is the whole block, [ix * num_y + iy]
    int num_x, num_y;

    Array2D(int Nx, int Ny, bool init_zero = true) : ptr(new T*[Nx]), num_x(Nx), num_y(Ny)
    {
        T* block = new T[static_cast<long>(Nx) * Ny];
        for (int ix = 0; ix < Nx; ++ix) ptr[ix] = block + static_cast<long>(ix) * Ny;
        if (init_zero) zero();
    }
    ~Array2D()
    {
        delete[] ptr[0];
        delete[] ptr;
    }
    Array2D(const Array2D&) = delete;
    Array2D& operator=(const Array2D&) = delete;

    T* operator[](int ix) { return ptr[ix]; }
    operator T**() { return ptr; }
    operator T*() { return ptr[0]; }
    int size_x() { return num_x; }
    int size_y() { return num_y; }
    long count() const { return static_cast<long>(num_x) * num_y; }
    void zero() { krhost::clear_all(ptr[0], count()); }

    void write(std::ofstream* out) { out->write(reinterpret_cast<char*>(ptr[0]), sizeof(T) * count()); }
    void read(std::ifstream* in) { in->read(reinterpret_cast<char*>(ptr[0]), sizeof(T) * count()); }

    template <typename U>
    void operator/=(Array2D<U>& o)
    {
        if (o.num_x != num_x || o.num_y != num_y) { std::cerr << "Array2D ERROR: Cannot divide arrays with different dimensions"; return; }
        krhost::divide_each(ptr[0], o.ptr[0], count());
    }
    template <typename U>
    void operator+=(Array2D<U>& o)
    {
        if (o.num_x != num_x || o.num_y != num_y) { std::cerr << "Array2D ERROR: Cannot add arrays with different dimensions"; return; }
        krhost::add_each(ptr[0], o.ptr[0], count());
    }
    KR_ARRAY_SCALAR_DIVISIONS(ptr[0], count())
};

template <typename T>
class Array3D {
public:
    T*** ptr;           // ptr[ix][iy] -> num_z elements
    T* pool;            // the whole block, [(ix * num_y + iy) * num_z + iz]
    int num_x, num_y, num_z;

    Array3D(int Nx, int Ny, int Nz, bool init_zero = true) : ptr(new T**[Nx]), pool(new T[static_cast<long>(Nx) * Ny * Nz]), num_x(Nx), num_y(Ny), num_z(Nz)
    {
        T** rows = new T*[static_cast<long>(Nx) * Ny];
        for (long r = 0; r < static_cast<long>(Nx) * Ny; ++r) rows[r] = pool + r * Nz;
        for (int ix = 0; ix < Nx; ++ix) ptr[ix] = rows + static_cast<long>(ix) * Ny;
        if (init_zero) zero();
    }
    ~Array3D()
    {
        delete[] ptr[0];
        delete[] ptr;
        delete[] pool;
    }
    Array3D(const Array3D&) = delete;
    Array3D& operator=(const Array3D&) = delete;

    T& elem(int i, int j, int k) { return ptr[i][j][k]; }
    T** operator[](int ix) { return ptr[ix]; }
    operator T**() { return ptr[0]; }
    operator T*() { return pool; }
    int size_x() { return num_x; }
    int size_y() { return num_y; }
    int size_z() { return num_z; }
    long count() const { return static_cast<long>(num_x) * num_y * num_z; }
    void zero() { krhost::clear_all(pool, count()); }

    void write(std::ofstream* out) { out->write(reinterpret_cast<char*>(pool), sizeof(T) * count()); }
    void read(std::ifstream* in) { in->read(reinterpret_cast<char*>(pool), sizeof(T) * count()); }

    template <typename U>
    void operator/=(Array3D<U>& o)
    {
        if (o.num_x != num_x || o.num_y != num_y || o.num_z != num_z) { std::cerr << "Array3D ERROR: Cannot divide arrays with different dimensions"; return; }
        krhost::divide_each(pool, o.pool, count());
    }
    template <typename U>
    void operator+=(Array3D<U>& o)
    {
        if (o.num_x != num_x || o.num_y != num_y || o.num_z != num_z) { std::cerr << "Array3D ERROR: Cannot add arrays with different dimensions"; return; }
        krhost::add_each(pool, o.pool, count());
    }
    KR_ARRAY_SCALAR_DIVISIONS(pool, count())
};

#endif /* ARRAY_H_ */
