#include "cvshim.hpp"
