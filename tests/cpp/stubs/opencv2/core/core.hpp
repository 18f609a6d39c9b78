// Declarations only (g++ -fsyntax-only of the drop-in wrappers); see tests/cpp/stubs/README.md.
#pragma once
#include <cstddef>
#include <vector>
typedef unsigned char uchar;
#define CV_8U 0
#define CV_8UC1 0
#define CV_Assert(x) ((void)(x))
namespace cv {
struct Point2f { float x, y; Point2f(); Point2f(float, float); };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
class MatExpr;
class Mat {
 public:
  Mat();
  Mat(int rows, int cols, int type, void* data);
  Mat(const MatExpr&);
  int rows, cols;
  uchar* data;
  size_t step;
  int type() const;
  bool empty() const;
  Mat clone() const;
  Mat rowRange(int, int) const;
  Mat colRange(int, int) const;
  Mat row(int) const;
  Mat col(int) const;
  MatExpr t() const;
  double dot(const Mat&) const;
  template <typename T> T* ptr(int i = 0);
  template <typename T> const T* ptr(int i = 0) const;
  template <typename T> T& at(int i);
  template <typename T> const T& at(int i) const;
  template <typename T> T& at(int i, int j);
  template <typename T> const T& at(int i, int j) const;
};
class MatExpr {
 public:
  operator Mat() const;
  MatExpr t() const;
};
MatExpr operator*(const Mat&, const Mat&);
MatExpr operator*(const MatExpr&, const Mat&);
MatExpr operator*(double, const Mat&);
MatExpr operator*(double, const MatExpr&);
MatExpr operator+(const MatExpr&, const Mat&);
MatExpr operator-(const Mat&, const Mat&);
MatExpr operator-(const Mat&);
MatExpr operator-(const MatExpr&);
MatExpr operator/(const Mat&, double);
double norm(const Mat&);
class _InputArray { public: _InputArray(const Mat&); bool empty() const; Mat getMat() const; };
class _OutputArray { public: _OutputArray(Mat&); void release() const; void create(int, int, int) const; Mat getMat() const; };
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
