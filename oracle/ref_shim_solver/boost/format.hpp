// stand-in for <boost/format.hpp>: include/teaser/utils.h of the reference includes it for commented-out file dumps
#pragma once
