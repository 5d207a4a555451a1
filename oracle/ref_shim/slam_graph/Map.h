// TEST INFRASTRUCTURE: the adapters include the reference header of this name; in the runnable graph build it is the stand-in (oracle/ref_shim/slam_graph_standins.hpp)
#include "../slam_graph_standins.hpp"
