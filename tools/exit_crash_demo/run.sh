# What a crash in a dependency's static destructor looks like from outside: the program has finished, printed its results
# and returned 0 from main -- and a parent that captures its stdout through a pipe sees NOTHING and a SIGSEGV exit status, because
# std::cout's buffer is flushed by the last ios_base::Init destructor, which runs after the faulting one.
#   bash tools/exit_crash_demo/run.sh   ->   stdout: "" ; exit status 139 (= 128 + SIGSEGV)
set -e
d=$(mktemp -d)
g++ -O2 -shared -fPIC "$(dirname "$0")/lib.cpp" -o $d/libboom.so
g++ -O2 "$(dirname "$0")/main.cpp" -o $d/m -L$d -lboom -Wl,-rpath,$d
set +e
out=$($d/m | cat; echo "exit status ${PIPESTATUS[0]}")
echo "captured: [$out]"
rm -rf $d
