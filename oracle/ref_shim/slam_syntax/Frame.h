// TEST INFRASTRUCTURE: the adapters include the reference header of this name; in the syntax check it is the stand-in declaration (slam_standins.hpp)
#include "slam_standins.hpp"
