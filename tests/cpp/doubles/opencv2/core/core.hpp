// Functional test double of the cv:: names the drop-in wrappers use; see tests/cpp/doubles/README.md.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>
typedef unsigned char uchar;
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(x) do { if (!(x)) throw std::runtime_error("CV_Assert: " #x); } while (0)
namespace cv {
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float a, float b) : x(a), y(b) {} };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };

// Dense 2-D matrix, CV_8U or CV_32F, row-major with a byte step; views share the owner's buffer.
class Mat {
 public:
  int rows, cols;
  uchar* data;
  size_t step;
  Mat() : rows(0), cols(0), data(nullptr), step(0), type_(CV_8U) {}
  Mat(int r, int c, int type) : rows(r), cols(c), step((size_t)c * esz(type)), type_(type) {
    buf_ = std::make_shared<std::vector<uchar> >((size_t)r * step + 16, 0);
    data = buf_->data();
  }
  Mat(int r, int c, int type, void* ext) : rows(r), cols(c), data((uchar*)ext), step((size_t)c * esz(type)), type_(type) {}
  int type() const { return type_; }
  bool empty() const { return rows == 0 || cols == 0 || !data; }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int i = 0; i < rows; i++) std::memcpy(m.data + (size_t)i * m.step, data + (size_t)i * step, (size_t)cols * esz(type_));
    return m;
  }
  Mat rowRange(int a, int b) const { Mat m(*this); m.rows = b - a; m.data = data + (size_t)a * step; return m; }
  Mat colRange(int a, int b) const { Mat m(*this); m.cols = b - a; m.data = data + (size_t)a * esz(type_); return m; }
  Mat row(int i) const { return rowRange(i, i + 1); }
  Mat col(int j) const { return colRange(j, j + 1); }
  Mat t() const {
    Mat m(cols, rows, type_);
    for (int i = 0; i < rows; i++)
      for (int j = 0; j < cols; j++) m.at<float>(j, i) = at<float>(i, j);
    return m;
  }
  double dot(const Mat& o) const {  // sum of element products, in double
    double s = 0;
    for (int i = 0; i < rows; i++)
      for (int j = 0; j < cols; j++) s += (double)at<float>(i, j) * (double)o.at<float>(i, j);
    return s;
  }
  template <typename T> T* ptr(int i = 0) { return reinterpret_cast<T*>(data + (size_t)i * step); }
  template <typename T> const T* ptr(int i = 0) const { return reinterpret_cast<const T*>(data + (size_t)i * step); }
  template <typename T> T& at(int i, int j) { return ptr<T>(i)[j]; }
  template <typename T> const T& at(int i, int j) const { return ptr<T>(i)[j]; }
  template <typename T> T& at(int i) { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }  // vectors
  template <typename T> const T& at(int i) const { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }

 private:
  static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
  int type_;
  std::shared_ptr<std::vector<uchar> > buf_;
};
typedef Mat MatExpr;

inline Mat operator*(const Mat& a, const Mat& b) {  // products accumulated in double, one rounding to float
  Mat m(a.rows, b.cols, CV_32F);
  for (int i = 0; i < a.rows; i++)
    for (int j = 0; j < b.cols; j++) {
      double s = 0;
      for (int k = 0; k < a.cols; k++) s += (double)a.at<float>(i, k) * (double)b.at<float>(k, j);
      m.at<float>(i, j) = (float)s;
    }
  return m;
}
inline Mat scaled(const Mat& a, double f) {
  Mat m(a.rows, a.cols, CV_32F);
  for (int i = 0; i < a.rows; i++)
    for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = (float)((double)a.at<float>(i, j) * f);
  return m;
}
inline Mat operator*(double f, const Mat& a) { return scaled(a, f); }
inline Mat operator/(const Mat& a, double f) {
  Mat m(a.rows, a.cols, CV_32F);
  for (int i = 0; i < a.rows; i++)
    for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = (float)((double)a.at<float>(i, j) / f);
  return m;
}
inline Mat operator+(const Mat& a, const Mat& b) {
  Mat m(a.rows, a.cols, CV_32F);
  for (int i = 0; i < a.rows; i++)
    for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = a.at<float>(i, j) + b.at<float>(i, j);
  return m;
}
inline Mat operator-(const Mat& a, const Mat& b) {
  Mat m(a.rows, a.cols, CV_32F);
  for (int i = 0; i < a.rows; i++)
    for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = a.at<float>(i, j) - b.at<float>(i, j);
  return m;
}
inline Mat operator-(const Mat& a) { return scaled(a, -1.0); }
inline double norm(const Mat& a) { return std::sqrt(a.dot(a)); }

class _InputArray {
 public:
  _InputArray(const Mat& m) : m_(m) {}
  bool empty() const { return m_.empty(); }
  Mat getMat() const { return m_; }
 private:
  Mat m_;
};
class _OutputArray {
 public:
  _OutputArray(Mat& m) : m_(&m) {}
  void release() const { *m_ = Mat(); }
  void create(int r, int c, int type) const { *m_ = Mat(r, c, type); }
  Mat getMat() const { return *m_; }
 private:
  Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
