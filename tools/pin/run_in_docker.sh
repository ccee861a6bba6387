#!/bin/bash
# One command for whoever has a network and docker: build and run the probe against the REAL Eigen / grid_map_core / tf2 /
# KDL / glibc of ROS Noetic, compare with the oracle, and leave tests/golden/pin_vectors_noetic.json behind.
#
#   tools/pin/run_in_docker.sh            (from anywhere inside the repository; needs docker and ~2 GB of image)
#
# What it does:  docker run ros:noetic-perception  ->  apt install ros-noetic-grid-map-core ros-noetic-tf2-geometry-msgs
#                g++ probe.cpp (catkin Release flags: -O2, no -march)  ->  ./probe > vectors.json
#                python3 compare.py vectors.json (on the host if python3 + numpy + gcc are there, else inside the container)
# Exit status = compare.py's: 0 = every third-party convention of the oracle is reproduced bit for bit by one of its variants;
# then commit tests/golden/pin_vectors_noetic.json -- tests/test_pin_kit_cpu.py::test_oracle_reproduces_pin_vectors (skipped while
# the file is absent) holds the oracle to it from then on, and "parity unpinned" can go from oracle/gg_oracle.h and DESIGN.md.
set -euo pipefail
root=$(cd "$(dirname "$0")/../.." && pwd)
image=${GG_PIN_IMAGE:-ros:noetic-perception}
out="$root/tests/golden/pin_vectors_noetic.json"
docker run --rm -v "$root":/repo -w /repo/tools/pin "$image" bash -c '
  set -e
  apt-get update -qq
  DEBIAN_FRONTEND=noninteractive apt-get install -y -qq ros-noetic-grid-map-core ros-noetic-tf2-geometry-msgs ros-noetic-tf2 liborocos-kdl-dev python3-numpy gcc g++ >/dev/null
  source /opt/ros/noetic/setup.bash
  g++ -O2 -std=c++14 probe.cpp -o /tmp/probe $(pkg-config --cflags eigen3) -I/opt/ros/noetic/include \
      -L/opt/ros/noetic/lib -lgrid_map_core -ltf2 -lorocos-kdl -lrostime -lcpp_common -lroscpp_serialization \
      -Wl,-rpath,/opt/ros/noetic/lib
  /tmp/probe > /repo/tests/golden/pin_vectors_noetic.json
  echo "probe: $(wc -c < /repo/tests/golden/pin_vectors_noetic.json) bytes of vectors written"
  dpkg -s libeigen3-dev ros-noetic-grid-map-core ros-noetic-tf2 liborocos-kdl-dev | grep -E "^(Package|Version)" | paste - - > /repo/tests/golden/pin_vectors_noetic.versions.txt
  cd /repo && python3 tools/pin/compare.py tests/golden/pin_vectors_noetic.json
'
rc=$?
echo "compare.py exit status $rc; vectors in $out (library versions next to it)"
exit $rc
