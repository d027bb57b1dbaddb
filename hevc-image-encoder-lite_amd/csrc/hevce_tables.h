/*
 * hevce_tables.h -- numeric tables of the HEVC specification shared by the device code and the host writer.
 *   LPS4     : H.265 table 9-46 rangeTabLps, four bytes per state packed little-endian   [ref HEVCe.c:704-713]
 *   LPSNEXT  : H.265 table 9-47 transIdxLps                                              [ref HEVCe.c:702]
 *   CTXINIT  : H.265 9.3.2.2 initValue for I slices, in the compact layout of hevce_common.h [ref HEVCe.c:763-777]
 *   LASTGRP / LASTMIN : last_sig_coeff prefix groups                                       [ref HEVCe.c:1047-1048]
 *   SIG4 / SIGPOS     : sig_coeff_flag context offsets                                      [ref HEVCe.c:1093-1094]
 */
#pragma once
#define HEVCE_TAB_LPS4 { \
    0xf0d0b080u, 0xe3c5a780u, 0xd8bb9e80u, 0xcdb2967bu, 0xc3a98e74u, 0xb9a0876fu, 0xaf988069u, 0xa6907a64u, \
    0x9e89745fu, 0x96826e5au, 0x8e7b6855u, 0x87756351u, 0x806f5e4du, 0x7a695949u, 0x74645545u, 0x6e5f5042u, \
    0x685a4c3eu, 0x6356483bu, 0x5e514538u, 0x594d4135u, 0x55493e33u, 0x50453b30u, 0x4c42382eu, 0x483f352bu, \
    0x453b3229u, 0x41383027u, 0x3e362d25u, 0x3b332b23u, 0x38302921u, 0x352e2720u, 0x322b251eu, 0x3029231du, \
    0x2d27211bu, 0x2b251f1au, 0x29231e18u, 0x27211c17u, 0x25201b16u, 0x231e1a15u, 0x211d1814u, 0x1f1b1713u, \
    0x1e1a1612u, 0x1c191511u, 0x1b171410u, 0x1916130fu, 0x1815120eu, 0x1714110eu, 0x1613100du, 0x15120f0cu, \
    0x14110e0cu, 0x13100e0bu, 0x120f0d0bu, 0x110f0c0au, 0x100e0c0au, 0x0f0d0b09u, 0x0e0c0b09u, 0x0e0c0a08u, \
    0x0d0b0908u, 0x0c0b0907u, 0x0c0a0907u, 0x0b0a0807u, 0x0b090806u, 0x0a090706u, 0x09080706u, 0x02020202u }
#define HEVCE_TAB_LPSNEXT { \
     0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9,11,11,12,13,13,15,15,16,16,18,18,19,19,21,21,22,22,23,24, \
    24,25,26,26,27,27,28,29,29,30,30,30,31,32,32,33,33,33,34,34,35,35,35,36,36,36,37,37,37,38,38,63 }
#define HEVCE_TAB_CTXINIT { \
    139, 141, 157, 184, 184, 63, 153, 138, 138, 111, 141, 94, \
    110, 110, 124, 0, 0, 125, 140, 153, 0, 0, 125, 127, 140, 109, 0, 111, 143, 127, 111, 79, \
    110, 110, 124, 0, 0, 125, 140, 153, 0, 0, 125, 127, 140, 109, 0, 111, 143, 127, 111, 79, \
    91, 171, \
    111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 0, \
    140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, \
    138, 153, 136, 167, 0, 0 }
#define HEVCE_TAB_LASTGRP { 0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8,8,8,8,8,8,8,8,9,9,9,9,9,9,9,9 }
#define HEVCE_TAB_LASTMIN { 0, 1, 2, 3, 4, 6, 8, 12, 16, 24 }
#define HEVCE_TAB_SIG4 { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 }
#define HEVCE_TAB_SIGPOS { 2, 1, 1, 0, 0, 0, 0, 0 }
