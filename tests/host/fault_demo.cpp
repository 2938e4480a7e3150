// A driver-shaped program (drivers/driver_common.hpp) that finishes its work and then dies in the static teardown of a library it links
// (tools/exit_crash_demo/lib.cpp): what it printed must reach a pipe, and the fault must name its phase and print a backtrace.
#include "../../quantum-mg_amd/drivers/driver_common.hpp"
extern "C" void touch();
int main() {
  qmg_driver::Guard guard;
  touch();
  qmg_driver::phase("solve");
  std::cout << "result line 1\n" << "result line 2\n";
  return qmg_driver::leave(0);
}
