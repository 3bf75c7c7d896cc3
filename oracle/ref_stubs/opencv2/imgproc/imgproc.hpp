#include "opencv2/features2d.hpp"
