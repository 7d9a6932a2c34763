# Sourced by build.sh when HSEFR_ASAN=1 (CPU only; listed in .gpurunignore -- see build.sh).  Host-only compile of the product's sources
# under AddressSanitizer + UndefinedBehaviorSanitizer: everything libhsefr does with a caller-supplied plan blob before it touches a device.
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
FLAGS="--offload-arch=gfx950 --cuda-host-only -O1 -g -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result -Wno-inline-asm $SAN"
LINK="-fsanitize=address,undefined"
SAN_EXE_FLAGS="--cuda-host-only -O1 -g -std=c++17 $SAN"
