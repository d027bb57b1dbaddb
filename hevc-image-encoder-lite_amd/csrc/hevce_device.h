/*
 * hevce_device.h -- the per-CTU analysis loop of an HEVC intra still-picture encoder, written for
 * CDNA4 (gfx950): ONE WORKGROUP (8 wavefronts of 64) WALKS ONE IMAGE'S CTUs in raster order with all
 * coder / neighbour state resident in LDS.  Bit-exact with lidongxuan/HEVC-image-encoder-lite
 * (reference src/HEVCe.c); each routine cites the reference lines whose results it reproduces.
 *
 * Why one workgroup per image: every RD decision is costed with the live CABAC state, so an image is a
 * single dependency chain.  Parallelism is taken (a) across images = workgroups,
 * (b) inside a CU: the 35 prediction modes x {one TU, four TUs, NxN} run as wave-level "TU jobs"
 * (lane = one column / row of a transform block) and as lane-level trial encodes (lane = one candidate).
 *
 * The code is split by role into the hevce_dev_*.h files included below, in dependency order (one translation unit: they share
 * macros and are not stand-alone headers).
 *
 * The same source is compiled by tests/emu (HEVCE_EMU) for a CPU fiber emulation used by the CPU test
 * suite to debug the control flow; that build is test infrastructure and is never part of the library.
 */
#pragma once
#include "hevce_common.h"
#include "hevce_tables.h"

#include "hevce_dev_base.h"
#include "hevce_dev_syntax.h"
#include "hevce_dev_tu.h"
#include "hevce_dev_nxn.h"
#include "hevce_dev_cu.h"
#include "hevce_dev_writer.h"
#include "hevce_dev_rounds.h"
#include "hevce_dev_cluster.h"
#include "hevce_dev_workers.h"
#include "hevce_dev_wrgroup.h"
#include "hevce_dev_picture.h"
