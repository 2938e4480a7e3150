// Prints two result lines with "\n" (no flush) and returns 0.  Linked against lib.cpp's library.
#include <iostream>
extern "C" void touch();
int main() { touch(); std::cout << "result line 1\n" << "result line 2\n"; return 0; }
