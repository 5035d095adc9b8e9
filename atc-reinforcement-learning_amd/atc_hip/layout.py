"""Word layout of the scenario blob and flag/mode bits — Python mirror of include/atc_step.h.

tests/test_abi.py parses the header and checks that every constant here matches it.
"""
ABI_VERSION = 21
BLOB_VERSION = 1014.0

H_VERSION, H_NWORDS, H_N_MVA, H_N_NOISE, H_N_ENTRY, H_OFF_POLY, H_OFF_VERT, H_OFF_ENTRY, H_OFF_GRID, H_N_VERTW = range(10)
H_OFF_SLOT = 10
H_OFF_SPAWN = 11
SPAWN_WORDS = 16
C_RWY_X, C_RWY_Y, C_RWY_H, C_PHI_TO_RWY, C_FAF_X, C_FAF_Y, C_NRM_X, C_NRM_Y = range(16, 24)
C_FAF_ANGLE, C_GS_TAN, C_FAF_MVA, C_WORLD_DIAG, C_NM_TO_FT = range(24, 29)
C_V_MIN, C_V_MAX, C_H_MIN, C_H_MAX, C_A_MIN, C_A_MAX, C_HDOT_MIN, C_HDOT_MAX = range(29, 37)
C_PHIDOT_MIN, C_PHIDOT_MAX, C_V_INIT = 37, 38, 39
C_TRI_H, C_TRI_1, C_TRI_2 = 40, 48, 56
C_NORM_MIN, C_NORM_MAX, C_ACT_DISCR, C_BBOX = 64, 74, 84, 87
C_DIR_RWY_X, C_DIR_RWY_Y = 91, 92
C_ALIGNED_OK = 93
C_POS_X0, C_POS_Y0, C_POS_SCALE, C_POS_INV = 94, 95, 100, 101   # fixed-point position grid: nm = X0 + fix * 2^-k
C_FAF_FIX = 124                                                 # FAF on the grid: x (hi, lo), y (hi, lo); fix = hi * 65536 + lo
POS_MAX_K = 27
# speed / heading state: 32-bit fixed point (include/atc_step.h, ABI 18; the heading field saturates since ABI 19: below): kt = v_fix 2^-23 (unsigned), deg = 180 + phi_fix 2^-23
V_FIX_SHIFT, PHI_FIX_SHIFT, PHI_FIX_OFFSET = 23, 23, 180.0
# ABI 19: the heading is unbounded like the reference's (model.py:104-120) — a 32-bit field that SATURATES (INT32_MIN / INT32_MAX =
# "WIDE") in front of the exact integer-valued float64 counts in atc_state_t.phi_wide[i][0] ([1]: the last heading target)
PHI_WIDE_WORDS = 4
PHI_LIMIT = 2.0 ** 52
I32_MIN, I32_MAX = -2 ** 31, 2 ** 31 - 1
C_TRI_BBOX = 96
C_NORM_A, C_NORM_B = 104, 114
C_END = 128

P_MINX, P_MINY, P_MAXX, P_MAXY, P_HEIGHT, P_VOFF, P_NVERT, P_PENALTY, P_WORDS = range(9)
E_X, E_Y, E_PHI, E_NLEV, E_LEV0 = range(5)
E_WORDS, E_MAXLEV = 12, 8
G_X0, G_Y0, G_INV, G_NX, G_NY, G_OFF_POOL, G_NREC = range(7)
G_HDR = 8
GE_WORDS = 8

PKT_CHUNKS = 9
MAX_AIRCRAFT = 64
OBS_DIM = 10
ACT_DIM = 3

F_BELOW_MVA, F_OUTSIDE, F_WON, F_TIMEOUT = 1, 2, 4, 8
F_INVALID_V, F_INVALID_H, F_CONFLICT, F_NOISE, F_INACTIVE, F_PHI_LIMIT = 16, 32, 64, 128, 256, 512
F_TERMINAL = F_BELOW_MVA | F_OUTSIDE | F_TIMEOUT | F_CONFLICT  # episode-ending on their own (WON: when all handed over)

M_REWARD_SHAPING, M_NORMALIZE, M_DISCRETE, M_AUTO_RESET, M_RANDOM_ENTRY, M_KEEP_ACTIVE, M_ACTIONS_HELD = 1, 2, 4, 8, 16, 32, 64

# per-step env record (atc_state_t.env): 4 x 32-bit words, float fields by bit pattern
ENV_TIMESTEPS, ENV_ACTIONS_TAKEN, ENV_TOTAL_REWARD, ENV_MASK_LO, ENV_WORDS = range(5)
# per-episode env record (atc_state_t.stats): 8 x 32-bit words
STAT_EPISODES, STAT_EP_LENGTH, STAT_EP_RETURN, STAT_WIN_BITS, STAT_EP_ACTIONS, STAT_MASK_HI = range(6)
STAT_WORDS = 8
# aircraft record (atc_state_t.ac, 4 x int32): x / y position counts, heading counts, speed counts; atc_state_t.alt: float64 [ft];
# last-action record (atc_state_t.last_act, 4 x int32): speed counts, heading counts, altitude target float64 in words 2..3
AC_X, AC_Y, AC_PHI, AC_V, AC_WORDS = range(5)
LA_V, LA_PHI, LA_H, LA_WORDS = 0, 1, 2, 4
# LDS-resident lookup table (include/atc_step.h: ATC_LDS_*)
LDS_MAGIC = 0x3154444C
(LDS_H_MAGIC, LDS_H_BYTES, LDS_H_X0, LDS_H_Y0, LDS_H_INV, LDS_H_NX, LDS_H_NY, LDS_H_OFF_L1, LDS_H_OFF_SUB, LDS_H_N_SUB, LDS_H_OFF_LINE,
 LDS_H_N_LINE, LDS_H_OFF_HTS, LDS_H_SUB, LDS_H_OFF_RESID, LDS_H_N_RESID, LDS_H_LDS_BYTES, LDS_H_OFF_POOL, LDS_H_N_REC) = range(19)
LDS_HDR_WORDS = 24
LDS_CLEAN, LDS_LINE, LDS_SUB, LDS_RESID = 0, 1, 2, 3
