#include "../opencv.hpp"
