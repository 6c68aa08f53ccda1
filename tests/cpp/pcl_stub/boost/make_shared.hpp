#pragma once
#include <utility>
#include "shared_ptr.hpp"
namespace boost {
template <typename T, typename... A>
shared_ptr<T> make_shared(A&&... a) { return shared_ptr<T>::from_std(std::make_shared<T>(std::forward<A>(a)...)); }
}  // namespace boost
