#pragma once
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
