// Stand-in for <flann/flann.hpp> (FLANN 1.9.1 is not installed): the interface the reference's feature_matcher.cc
// uses — flann::Matrix, flann::Index<flann::L2<float>>, KDTreeSingleIndexParams, SearchParams, buildIndex, knnSearch —
// over a brute-force scan.  KDTreeSingleIndex with eps = 0 is an EXACT search, so only two things of FLANN matter for
// the result: the distance functor (published flann::L2<float>::operator(): groups of four differences,
// result += d0*d0 + d1*d1 + d2*d2 + d3*d3, then the tail one by one) and which of several equidistant points is
// returned (tree-order dependent in FLANN; lowest index here — the oracle's declared divergence D3).
// Test infrastructure, not product code.
#pragma once
#include <cstddef>
#include <vector>
namespace flann {
template <typename T>
class Matrix {
 public:
  Matrix() : rows(0), cols(0), data_(nullptr) {}
  Matrix(T* data, size_t r, size_t c) : rows(r), cols(c), data_(data) {}
  T* operator[](size_t r) const { return data_ + r * cols; }
  T* ptr() const { return data_; }
  size_t rows, cols;

 private:
  T* data_;
};
template <typename T>
struct L2 {
  typedef T ElementType;
  typedef T ResultType;
  template <typename It1, typename It2>
  ResultType operator()(It1 a, It2 b, size_t size, ResultType /*worst_dist*/ = -1) const {
    ResultType result = ResultType();
    ResultType diff0, diff1, diff2, diff3;
    It1 last = a + size;
    It1 lastgroup = last - 3;
    while (a < lastgroup) {
      diff0 = (ResultType)(a[0] - b[0]);
      diff1 = (ResultType)(a[1] - b[1]);
      diff2 = (ResultType)(a[2] - b[2]);
      diff3 = (ResultType)(a[3] - b[3]);
      result += diff0 * diff0 + diff1 * diff1 + diff2 * diff2 + diff3 * diff3;
      a += 4;
      b += 4;
    }
    while (a < last) {
      diff0 = (ResultType)(*a++ - *b++);
      result += diff0 * diff0;
    }
    return result;
  }
};
struct KDTreeSingleIndexParams {
  explicit KDTreeSingleIndexParams(int leaf_max_size = 10) : leaf(leaf_max_size) {}
  int leaf;
};
struct SearchParams {
  explicit SearchParams(int checks_ = 32) : checks(checks_) {}
  int checks;
};
template <typename Distance>
class Index {
 public:
  typedef typename Distance::ElementType ElementType;
  typedef typename Distance::ResultType DistanceType;
  Index(const KDTreeSingleIndexParams& = KDTreeSingleIndexParams()) {}
  Index(const Matrix<ElementType>& pts, const KDTreeSingleIndexParams& = KDTreeSingleIndexParams()) : src_(pts) {}
  void buildIndex() {  // KDTreeSingleIndex copies nothing by default either, but the reference's dataset buffer dies with
                       // buildKDTree's scope (it relies on the reorder copy): keep our own copy
    data_.assign(src_.ptr(), src_.ptr() + src_.rows * src_.cols);
    rows_ = src_.rows;
    cols_ = src_.cols;
  }
  int knnSearch(const Matrix<ElementType>& queries, Matrix<int>& indices, Matrix<DistanceType>& dists, size_t knn,
                const SearchParams&) const {
    Distance d;
    for (size_t q = 0; q < queries.rows; ++q) {
      DistanceType best = 0;
      int bi = -1;
      for (size_t i = 0; i < rows_; ++i) {
        const DistanceType v = d(queries[q], data_.data() + i * cols_, cols_);
        if (bi < 0 || v < best) {
          best = v;
          bi = (int)i;
        }
      }
      for (size_t k = 0; k < knn; ++k) {
        indices[q][k] = bi;
        dists[q][k] = best;
      }
    }
    return (int)queries.rows;
  }

 private:
  Matrix<ElementType> src_;
  std::vector<ElementType> data_;
  size_t rows_ = 0, cols_ = 0;
};
}  // namespace flann
