// Stand-in for boost::shared_ptr: a distinct class template (NOT an alias of std::shared_ptr), so that code mixing the two
// does not compile — exactly as with the real Boost.
#pragma once
#include <cstddef>
#include <memory>
#include <type_traits>
namespace boost {
template <typename T>
class shared_ptr {
 public:
  using element_type = T;
  shared_ptr() = default;
  shared_ptr(std::nullptr_t) {}
  template <typename U, typename = typename std::enable_if<std::is_convertible<U*, T*>::value>::type>
  explicit shared_ptr(U* p) : p_(p) {}
  template <typename U, typename = typename std::enable_if<std::is_convertible<U*, T*>::value>::type>
  shared_ptr(const shared_ptr<U>& o) : p_(o.std_ptr()) {}
  T* get() const { return p_.get(); }
  T& operator*() const { return *p_; }
  T* operator->() const { return p_.get(); }
  explicit operator bool() const { return static_cast<bool>(p_); }
  void reset() { p_.reset(); }
  template <typename U>
  void reset(U* p) { p_.reset(p); }
  long use_count() const { return p_.use_count(); }
  const std::shared_ptr<T>& std_ptr() const { return p_; }
  static shared_ptr from_std(std::shared_ptr<T> p) {
    shared_ptr r;
    r.p_ = std::move(p);
    return r;
  }

 private:
  std::shared_ptr<T> p_;
};
template <typename T, typename U>
bool operator==(const shared_ptr<T>& a, const shared_ptr<U>& b) { return a.get() == b.get(); }
template <typename T, typename U>
bool operator!=(const shared_ptr<T>& a, const shared_ptr<U>& b) { return a.get() != b.get(); }
template <typename T>
bool operator==(const shared_ptr<T>& a, std::nullptr_t) { return a.get() == nullptr; }
template <typename T>
bool operator!=(const shared_ptr<T>& a, std::nullptr_t) { return a.get() != nullptr; }
}  // namespace boost
