// A shared library whose static object faults in its destructor -- the stand-in for a runtime library that crashes in its own
// teardown after main() has returned.  See run.sh.
#include <csignal>
#include <iostream>
struct Boom { ~Boom() { raise(SIGSEGV); } };
static Boom boom;
extern "C" void touch() {}
