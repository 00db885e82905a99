// opv_offset_host.h — host half of the offset search's tie decision (opv_offset_host.cpp; called from opv_capi.hip only)
#pragma once
#include <stddef.h>
#include <stdint.h>

struct OpvTieSlot;   // opv_device.h

// energy of ONE candidate the reference's way (src/opv-demod.cpp:143-159) with the host's libm; threads: share the windows out
double opv_offset_candidate_energy(const int16_t* iq, size_t nsym, double offset_hz, bool threads);
// the whole coarse + fine decision for one stream from the device's polynomial and input power, contenders re-evaluated on
// the host; fills the 134 energies in scan order and the number of re-evaluations, returns the estimate in Hz
double opv_offset_decide_on_host(const int16_t* iq, size_t nsym, const double* poly19, double power, double* energies134, uint32_t* ties_out,
                                 bool threads);
// the same for the n filled slots of one pass of the stream-ordered decision (results left in the slots); no HIP call inside
void opv_offset_decide_slots(OpvTieSlot* slots, uint32_t n);
// one-time probe: this process's sin / cos give the pinned energy below for the probe sequence
bool opv_offset_host_libm_matches_reference();

#define OPV_OFFSET_PROBE_HZ 1425.0
#define OPV_OFFSET_PROBE_ENERGY 0x1.a727112f5a965p+37   /* = entry 117 (+1425 Hz) of the reference search's energy table for the probe sequence (tests/test_capi_and_host.py) */
