#pragma once
#include <memory>
namespace boost {
template <typename T>
using shared_ptr = std::shared_ptr<T>;
}
