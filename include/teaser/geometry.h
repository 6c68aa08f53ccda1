// teaser/geometry.h — drop-in for url-kaist/Quatro's include/teaser/geometry.h: teaser::PointXYZ (three floats)
// and the std::vector-backed teaser::PointCloud container (:15-70) that the front-end classes take.
#pragma once
#include <cstddef>
#include <vector>

namespace teaser {

struct PointXYZ {
  float x, y, z;
  friend inline bool operator==(const PointXYZ& a, const PointXYZ& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
  friend inline bool operator!=(const PointXYZ& a, const PointXYZ& b) { return !(a == b); }
};

class PointCloud {
 public:
  using value_type = PointXYZ;
  using reference = PointXYZ&;
  using const_reference = const PointXYZ&;
  using size_type = std::vector<PointXYZ>::size_type;
  using difference_type = std::vector<PointXYZ>::difference_type;
  using iterator = std::vector<PointXYZ>::iterator;
  using const_iterator = std::vector<PointXYZ>::const_iterator;

  PointCloud() = default;
  iterator begin() { return pts_.begin(); }
  iterator end() { return pts_.end(); }
  const_iterator begin() const { return pts_.begin(); }
  const_iterator end() const { return pts_.end(); }
  std::size_t size() const { return pts_.size(); }
  void reserve(std::size_t n) { pts_.reserve(n); }
  bool empty() const { return pts_.empty(); }
  PointXYZ& operator[](std::size_t i) { return pts_[i]; }
  const PointXYZ& operator[](std::size_t i) const { return pts_[i]; }
  PointXYZ& at(std::size_t i) { return pts_.at(i); }
  const PointXYZ& at(std::size_t i) const { return pts_.at(i); }
  PointXYZ& front() { return pts_.front(); }
  const PointXYZ& front() const { return pts_.front(); }
  PointXYZ& back() { return pts_.back(); }
  const PointXYZ& back() const { return pts_.back(); }
  void push_back(const PointXYZ& p) { pts_.push_back(p); }
  void clear() { pts_.clear(); }

 private:
  std::vector<PointXYZ> pts_;
};

}  // namespace teaser
