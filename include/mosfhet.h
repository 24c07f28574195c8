/*
 * mosfhet.h -- the name MOSFHET programs include (test/benchmark.c:1, applications/leveled_lut/vertical_packing.c:6 of the reference).
 *
 * This is NOT the reference's header: it forwards to include/mosfhet_compat.h, the MI355X engine's restatement of the part of that API that
 * lies on the bootstrap path and its callers (every prototype there cites the reference line it replaces), and pulls in
 * the standard headers the reference's own header exposes to its includers (its lines 5-12), so that a program written against the
 * reference for this path compiles and links against libmosfhet_hip.so unchanged:
 *
 *     gcc -I<repo>/include app.c -L<repo>/mosfhet_amd -lmosfhet_hip -lm
 *
 * tests/test_host_and_abi.py builds the reference's own applications/leveled_lut/vertical_packing.c this way (where /root/reference exists).
 */
#ifndef MOSFHET_H_FORWARD
#define MOSFHET_H_FORWARD
#include <assert.h>
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mosfhet_compat.h"
#endif
