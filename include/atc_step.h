/*
 * atc_step.h — C-ABI of libatcstep.so, the MI355X (gfx950) implementation of the batched
 * AtcGym.step() hot path of fvalka/atc-reinforcement-learning.
 *
 * The reference has no FFI (pure Python).  Each entry point below replaces a Python call site of the
 * reference (cited as path:line relative to the reference tree) and is what a ctypes binding on the
 * reference side would load (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  No exceptions, no Python/torch objects cross the boundary.
 *   - every int-returning function: 0 = ok, ATC_ERR_* (<0) otherwise; atc_last_error() gives a
 *     thread-local message.
 *   - every buffer is a BORROWED device pointer (e.g. torch tensor .data_ptr()); the library owns only
 *     the opaque scenario handle.
 *   - all launches are asynchronous on the caller's HIP stream (`stream` = hipStream_t, e.g.
 *     torch.cuda.current_stream().cuda_stream); the caller synchronises.
 *   - aircraft arrays are SoA, index = env * N + k   (k = aircraft slot in the env, N <= 64).
 *   - re-entrant: no global mutable state besides the thread-local error string.
 */
#ifndef ATC_STEP_H
#define ATC_STEP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ATC_ABI_VERSION 21

/* ---------------------------------------------------------------------------------------------
 * Scenario blob: one flat array of 32-bit floats (device copy) compiled on the host from the sector
 * description (reference: envs/atc/scenarios.py:14-207, envs/atc/model.py:148-186,260-306,
 * envs/atc/atc_gym.py:45-58,88-110).  Integer fields are stored as exactly representable floats.
 * The float64 master (used by the f64 oracle) has the identical word layout.
 * ------------------------------------------------------------------------------------------- */
#define ATC_BLOB_VERSION 1014.0f
enum {
    ATC_H_VERSION = 0,   /* ATC_BLOB_VERSION */
    ATC_H_NWORDS = 1,    /* total words */
    ATC_H_N_MVA = 2,     /* number of MVA polygons (list order = lookup priority, model.py:283) */
    ATC_H_N_NOISE = 3,   /* number of noise-abatement polygons (extension; 0 for reference scenarios) */
    ATC_H_N_ENTRY = 4,   /* number of entry points (scenarios.py:192-207) */
    ATC_H_OFF_POLY = 5,  /* word offset of polygon table: MVA polygons first, then noise polygons */
    ATC_H_OFF_VERT = 6,  /* word offset of vertex pool (x,y interleaved; rings closed: first == last) */
    ATC_H_OFF_ENTRY = 7, /* word offset of entry-point table */
    ATC_H_OFF_GRID = 8,  /* word offset of the MVA lookup grid (0 = absent) */
    ATC_H_N_VERTW = 9,   /* words in the vertex pool */
    ATC_H_OFF_SLOT = 10, /* word offset of the slot-lattice spawn table: 64 x (x, y, phi, h) for aircraft slot k =
                            entry k mod n_entry at level (k div n_entry) mod n_levels (the non-random reset) */
    ATC_H_OFF_SPAWN = 11,/* word offset (64-byte aligned) of the device's SPAWN RECORDS, ATC_SPAWN_WORDS words each: 64 records
                            for the slot lattice (record k = aircraft slot k), then one per entry point (random resets; the
                            altitude of those comes from the drawn level).  A record is what AtcGym.reset computes for an
                            aircraft placed there (atc_gym.py:346-351,365), evaluated once on the host:
                              words 0..3   x_fix, y_fix (position grid counts), h (float), phi_fix (heading counts) — 32-bit
                                           PATTERNS in the fp32 device blob (integers are not stored as float values here)
                              words 4..13  the RAW reset observation _get_state(mva = 0), atc_gym.py:262-277,351
                              words 14..15 0
                            The float64 master holds the same quantities as values (nothing reads them there). */
    /* constants block */
    ATC_C_RWY_X = 16, ATC_C_RWY_Y = 17, ATC_C_RWY_H = 18,
    ATC_C_PHI_TO_RWY = 19,                /* (phi_from_runway + 180) % 360, model.py:163,245 */
    ATC_C_FAF_X = 20, ATC_C_FAF_Y = 21,   /* model.py:171-172 */
    ATC_C_NRM_X = 22, ATC_C_NRM_Y = 23,   /* _faf_iaf_normal, model.py:164 */
    ATC_C_FAF_ANGLE = 24,                 /* 45, model.py:167 */
    ATC_C_GS_TAN = 25,                    /* tan(3*pi/180), model.py:204 */
    ATC_C_FAF_MVA = 26,                   /* atc_gym.py:49 */
    ATC_C_WORLD_DIAG = 27,                /* atc_gym.py:58 */
    ATC_C_NM_TO_FT = 28,                  /* 6076, model.py:10 */
    ATC_C_V_MIN = 29, ATC_C_V_MAX = 30, ATC_C_H_MIN = 31, ATC_C_H_MAX = 32,          /* model.py:13 */
    ATC_C_A_MIN = 33, ATC_C_A_MAX = 34, ATC_C_HDOT_MIN = 35, ATC_C_HDOT_MAX = 36,     /* model.py:45-48 */
    ATC_C_PHIDOT_MIN = 37, ATC_C_PHIDOT_MAX = 38, ATC_C_V_INIT = 39,                  /* model.py:49-50; atc_gym.py:348 */
    ATC_C_TRI_H = 40,   /* corridor_horizontal ring [faf, corner1, corner2, faf], 8 words, model.py:177 */
    ATC_C_TRI_1 = 48,   /* corridor1 ring [faf, corner1, iaf, faf], model.py:180 */
    ATC_C_TRI_2 = 56,   /* corridor2 ring [faf, corner2, iaf, faf], model.py:181 */
    ATC_C_NORM_MIN = 64,  /* 10 words, atc_gym.py:88-98 */
    ATC_C_NORM_MAX = 74,  /* 10 words, atc_gym.py:99-110 */
    ATC_C_ACT_DISCR = 84, /* 3 words (5, 50, 0.5), atc_gym.py:84 */
    ATC_C_BBOX = 87,      /* x0,y0,x1,y1, model.py:294-301 */
    ATC_C_DIR_RWY_X = 91, ATC_C_DIR_RWY_Y = 92, /* rot_matrix(phi_to_runway) . [0,1], model.py:219 */
    ATC_C_ALIGNED_OK = 93, /* 1 if the reference's angle window (model.py:216-229) accepts phi == phi_to_runway exactly,
                              evaluated on the host with the reference's own expression (np.dot / arccos) */
    /* fixed-point position grid of the fp32 path (see "Aircraft positions" below): nm = X0 + fix * 2^-k */
    ATC_C_POS_X0 = 94, ATC_C_POS_Y0 = 95, /* grid origin [nm], integer-valued */
    ATC_C_POS_SCALE = 100,                /* 2^k  (counts per nm) */
    ATC_C_POS_INV = 101,                  /* 2^-k (nm per count) */
    ATC_C_TRI_BBOX = 96,  /* x0,y0,x1,y1 of the corridor_horizontal triangle (exact early-out for model.py:198) */
    ATC_C_NORM_A = 104,   /* 10 words: 1 / (0.5 max)              — normalisation (atc_gym.py:187-189) as one fma:  */
    ATC_C_NORM_B = 114,   /* 10 words: -(min + 0.5 max) / (0.5 max)                  obs = raw * A + B (float32) */
    ATC_C_FAF_FIX = 124,  /* 4 words: the FAF on the position grid, x then y, each as (hi, lo) with fix = hi * 65536 + lo,
                             lo in [0, 65536) — two exactly representable floats per 32-bit integer */
    ATC_C_END = 128
};
/* polygon table record (8 words) */
enum { ATC_P_MINX = 0, ATC_P_MINY = 1, ATC_P_MAXX = 2, ATC_P_MAXY = 3,
       ATC_P_HEIGHT = 4,  /* MVA height [ft] (model.py:265) or noise-area ceiling [ft] */
       ATC_P_VOFF = 5,    /* word offset of first vertex */
       ATC_P_NVERT = 6,   /* number of vertices in the closed ring (= len(area_as_list)) */
       ATC_P_PENALTY = 7, /* noise-area per-step penalty (0 for MVA polygons) */
       ATC_P_WORDS = 8 };
/* entry-point record (12 words): x, y, phi, n_levels, levels[8] (flight levels, x100 ft; model.py:309-315) */
#define ATC_SPAWN_WORDS 16
enum { ATC_E_X = 0, ATC_E_Y = 1, ATC_E_PHI = 2, ATC_E_NLEV = 3, ATC_E_LEV0 = 4, ATC_E_WORDS = 12, ATC_E_MAXLEV = 8 };
/* lookup grid (optional acceleration structure for Airspace.find_mva, model.py:282-289; results identical to the
 * ordered polygon scan by construction — see atc_hip/scenario.py:build_grid).  16-byte aligned in the blob.
 *   header 8 words : x0, y0, 1/cell, nx, ny, offset of the edge pool (from grid start), number of edge records, 0
 *   cells  ny*nx*2 : (c, v) with |c| = code + 64 * noise mask (bit q: the bounds of noise-abatement area q meet the cell)
 *                    + 2^22 if the bounds of the corridor's horizontal triangle (ATC_C_TRI_BBOX) meet the cell
 *                    + 2^23 (ATC_G_CELL_LINE) if the cell is a SPLIT cell, see below
 *                    c > 0  : dirty cell, code = n_records (< 64), v = first record: walk that many edge records
 *                    c <= 0 : clean cell, code = polygon + 1 and v = MVA height — every point has this answer;
 *                             code = 0: outside the airspace
 *                    The OUTERMOST ring of cells is (0, 0) — clean, outside, no noise candidate: a point beyond the grid is
 *                    answered by the border cell its clamped index names (atc_scenario_create refuses other grids).
 *   pool           : 8-word records, two 16-byte halves G | M (flags and folding rules: atc_hip/scenario.py:build_grid):
 *                    edge       G = p1x, p1y, p2x, p2y          M = min(p1y,p2y), max(p1y,p2y), polygon height, code
 *                    terminator G = polygon bounds x0,y0,x1,y1  M = 0, 0, polygon height, code
 *                    code = 16 * polygon index + flags                                                                */
enum { ATC_G_X0 = 0, ATC_G_Y0 = 1, ATC_G_INV = 2, ATC_G_NX = 3, ATC_G_NY = 4, ATC_G_OFF_POOL = 5, ATC_G_NREC = 6,
       ATC_G_HDR = 8, ATC_GE_WORDS = 8 };
#define ATC_G_CELL_LINE (1u << 23) /* |c| flag of a SPLIT cell: its first record (not counted in `code`) is a LINE record
                                      G = p1x, p1y, dx/dy, margin   M = left polygon + 1, left height, right polygon + 1, right height:
                                      xl = p1x + (y - p1y) dx/dy;  x < xl - margin -> the left answer, x > xl + margin -> the right
                                      one (polygon + 1 = 0: outside the airspace), else walk the `code` ordinary records behind it */
#define ATC_GE_TERM 1    /* terminator: (crossing parity xor BASE) and the bounds test (model.py:286-287) decide now */
#define ATC_GE_CERTAIN 2 /* the cell lies entirely left of this edge: crossing iff the two y tests pass */
#define ATC_GE_LAST 4    /* last edge of a polygon whose bounds contain the whole cell: parity xor BASE decides now */
#define ATC_GE_BASE 8    /* an odd number of the polygon's edges is crossed by EVERY point of the cell (not listed) */

/* ---------------------------------------------------------------------------------------------
 * LDS-resident lookup table (ABI 21, optional; atc_scenario_attach_lds_table).  A second, compact form of Airspace.find_mva
 * (model.py:282-289) for the multi-step launches of ONE-aircraft envs that fit one workgroup per CU (65 536 x 1): each workgroup
 * stages the table in LDS once per launch and a step then answers from LDS instead of gathering a lookup-grid cell from global
 * memory (one wavefront per SIMD: that gather and the records behind it are 1.2 of the step's 2.9 us).  Same answers as the
 * ordered polygon scan, by the construction of the lookup grid (atc_hip/scenario.py:build_lds_table):
 *   bytes [0, 96)  : 24 header words, ATC_LDS_H_* (x0, y0, 1 / cell as fp32 patterns; offsets in bytes from the table's start)
 *   -- staged in LDS (bytes [0, ATC_LDS_H_LDS_BYTES)) --
 *   level 1        : uint16 codes [ny][nx], cells of 1 / inv nm (0.5) over the padded sector; the outermost ring is 0
 *   level 2        : uint16 codes [n_sub][8][8], the sub-cells of every level-1 cell of kind SUB
 *   LINE records   : float [n_line][8] = p1x, p1y, dx/dy, margin | left polygon + 1, height, right polygon + 1, height (ATC_G_CELL_LINE)
 *   heights        : float [64], index polygon + 1 ([0] = 0)
 *   walk words     : uint32 [n_resid] = first record | n_records << 24 of a RESID sub-cell's edge records in the pool
 *   -- read from global memory --
 *   pool           : float [n_rec][8], edge / terminator records in the lookup grid's format ("pool" above)
 *   code           : bit 15 = the cell meets the bounds of the corridor's horizontal triangle (level 1 only); bits 14..13 = kind
 *                    (ATC_LDS_CLEAN: payload = polygon + 1, 0 = outside; ATC_LDS_LINE: payload = LINE record; ATC_LDS_SUB, level 1
 *                    only: payload = sub-cell block; ATC_LDS_RESID, level 2 only: a vertex or a second border inside the sub-cell,
 *                    payload = walk word); bits 12..0 = payload
 * A lane in a RESID sub-cell walks its records (ONE trip to global memory for the lanes concerned); a point inside a LINE record's
 * margin band sends its whole wavefront to the lookup grid for that step, so the table is only attached to a scenario that HAS a
 * grid — and has no noise-abatement areas (the codes carry no candidate masks).  Results are identical with and without the table.
 * ------------------------------------------------------------------------------------------- */
#define ATC_LDS_MAGIC 0x3154444Cu /* "LDT1" */
enum { ATC_LDS_H_MAGIC = 0, ATC_LDS_H_BYTES = 1, ATC_LDS_H_X0 = 2, ATC_LDS_H_Y0 = 3, ATC_LDS_H_INV = 4, ATC_LDS_H_NX = 5,
       ATC_LDS_H_NY = 6, ATC_LDS_H_OFF_L1 = 7, ATC_LDS_H_OFF_SUB = 8, ATC_LDS_H_N_SUB = 9, ATC_LDS_H_OFF_LINE = 10,
       ATC_LDS_H_N_LINE = 11, ATC_LDS_H_OFF_HTS = 12, ATC_LDS_H_SUB = 13, ATC_LDS_H_OFF_RESID = 14, ATC_LDS_H_N_RESID = 15,
       ATC_LDS_H_LDS_BYTES = 16, ATC_LDS_H_OFF_POOL = 17, ATC_LDS_H_N_REC = 18, ATC_LDS_HDR_WORDS = 24 };
enum { ATC_LDS_CLEAN = 0, ATC_LDS_LINE = 1, ATC_LDS_SUB = 2, ATC_LDS_RESID = 3 };

#define ATC_MAX_AIRCRAFT 64
#define ATC_OBS_DIM 10 /* atc_gym.py:262-277 */
#define ATC_ACT_DIM 3  /* v, h, phi  (atc_gym.py:70,79) */

/* per-aircraft flag word (outputs `flags`, 16 bits) */
enum {
    ATC_F_BELOW_MVA = 1u << 0, /* atc_gym.py:149-153 */
    ATC_F_OUTSIDE = 1u << 1,   /* atc_gym.py:156-161 */
    ATC_F_WON = 1u << 2,       /* atc_gym.py:163-169 */
    ATC_F_TIMEOUT = 1u << 3,   /* atc_gym.py:171-173 */
    ATC_F_INVALID_V = 1u << 4, /* atc_gym.py:312-315 via model.py:69-72 */
    ATC_F_INVALID_H = 1u << 5, /* atc_gym.py:312-315 via model.py:91-94 */
    ATC_F_CONFLICT = 1u << 6,  /* extension: 3 nm / 1000 ft separation lost (README.md:51) */
    ATC_F_NOISE = 1u << 7,     /* extension: inside a noise-abatement area below its ceiling (README.md:62) */
    ATC_F_INACTIVE = 1u << 8,  /* extension: aircraft already handed over (won earlier in this episode) */
    ATC_F_PHI_LIMIT = 1u << 9  /* the heading target of this step lay beyond +-2^52 counts (|a_phi| > 2.98e6) and was clamped there:
                                  the format's one remaining bound on the reference's unvalidated heading (model.py:104-120) */
};

/* params.mode bits */
enum {
    ATC_M_REWARD_SHAPING = 1u << 0,  /* SimParameters.reward_shaping, model.py:143 */
    ATC_M_NORMALIZE = 1u << 1,       /* SimParameters.normalize_state, model.py:144 */
    ATC_M_DISCRETE = 1u << 2,        /* SimParameters.discrete_action_space, model.py:145 */
    ATC_M_AUTO_RESET = 1u << 3,      /* VecEnv semantics: done envs are reset inside the step, obs := raw reset obs */
    ATC_M_RANDOM_ENTRY = 1u << 4,    /* reset draws (entry, level) from the counter-based RNG; else slot lattice */
    ATC_M_KEEP_ACTIVE = 1u << 5,     /* the reference's single-aircraft rule (atc_gym.py:163-169): an aircraft that reaches the
                                        corridor ends the episode and STAYS under control (no hand-over), so stepping on
                                        without reset keeps simulating it (learning/atc-gym-compute-performance.py:14-16) */
    ATC_M_ACTIONS_HELD = 1u << 6     /* atc_step only — a promise of the caller: `actions` holds, for every aircraft, the same
                                        bits as in the previous step of these envs — an atc_step or the last step of a multi-step launch (frame skip,
                                        learning/atc-gym-demo.py:18-19).  Results are identical to a launch without the bit;
                                        the kernel then skips the last_action record (16 of 118 bytes per aircraft-step):
                                        an aircraft under control in the previous step has last_action == its accepted
                                        targets (atc_gym.py:305-311), so no action is counted and the record does not
                                        change.  Envs reset since their last step (timesteps == 0) are handled in full.
                                        Set on a launch whose actions DID change, actions_taken / last_action are wrong. */
};

typedef struct atc_params {
    double dt;              /* SimParameters.timestep [s], model.py:132-145 — a Python float in the reference: float64 here too
                               (ABI 20; 0.1 is not 0.1f), it scales every rate limit (model.py:75-78,97-100,117-120), the
                               displacement (model.py:126) and the base reward (atc_gym.py:137) */
    int32_t timestep_limit; /* 6000, atc_gym.py:40 */
    uint32_t mode;          /* ATC_M_* */
    uint64_t seed;          /* RNG key for ATC_M_RANDOM_ENTRY */
    float sep_nm;           /* 3.0  (extension) */
    float sep_ft;           /* 1000 (extension) */
    float conflict_reward;  /* -200 (extension) */
    uint32_t reserved0;     /* tag written into every chunk of atc_out_t.packet (ignored without a packet) */
    float reserved1;
    uint32_t reserved2;
} atc_params_t;

/* Aircraft positions (model.py:33-34) on the fp32 path.
 * The reference accumulates x, y in float64.  An fp32 accumulator drifts by up to half an ulp PER STEP on a straight leg
 * (1e-4 nm after 100 steps, measured) — beyond the 1e-5 bar.  Positions are therefore kept as 32-bit FIXED POINT on a
 * per-sector grid: nm = origin + fix * 2^-k, origin integer-valued, k the largest exponent (<= 27) whose range
 * +-2^(31-k) nm covers 1.5 x the sector's half extent (LOWW: origin (36, 42), k = 25: 3e-8 nm = 55 um steps, +-64 nm).
 *   - the per-step displacement (model.py:122-129; float64 from the fixed-point speed and heading, see below) advances the
 *     position by its value in counts rounded ONCE, saturating;
 *   - the fp32 position every formula of the reference sees is  (float)(origin + fix * 2^-k)  — ONE rounding (evaluated
 *     in float64, exact before the final conversion);
 *   - differences of positions are exact integers: the vector to the FAF (atc_gym.py:289-297) is
 *     (float)(faf_fix - fix) * 2^-k, accurate to fp32 RELATIVE precision however close the aircraft is to the FAF.
 * Accumulated rounding over a 6 000-step episode: <= 1.04e-6 nm measured against the float64 reference (dithered rounding, below);
 * half the bytes of a float64 pair.  An aircraft that is flown on, without reset, beyond the grid range (>= 24 nm outside
 * the LOWW bounding box) is pinned at the range limit: it stays "outside the airspace" exactly like the reference's
 * (model.py:289), only its x / y observation stops growing.
 *
 * Speed and heading (model.py:35-37, 60-129) on the fp32 path — ABI 18; unbounded heading: ABI 19.
 * The reference holds v, phi and the decoded action targets in float64.  As fp32 values they carry the rounding of the
 * target (1.5e-5 deg at 340 deg, 1.5e-5 kt at 250 kt) for as long as the target is held, and a heading that is 1e-5 deg off
 * moves the aircraft 5e-6 nm off over a 30 nm leg: next to the FAF, where the bearing to it is ill-conditioned, that was the one
 * stated exception to the 1e-5 bar of rounds 1-3 (tools/faf_conditioning.py).  Both are therefore 32-bit FIXED POINT, like
 * the positions, and the step's displacement is evaluated in float64 from them:
 *   speed    kt  = v_fix * 2^-23            UNSIGNED counts, [0, 512): the aircraft's [100, 300] and every refusable target
 *   heading  deg = 180 + P * 2^-23          signed counts P.  The reference never validates or wraps a heading (model.py:104-120:
 *                                           a continuous action a_phi = 3 turns the aircraft on to 720 deg), so P is NOT a 32-bit
 *                                           quantity: |P| <= 2^52 (+-5.4e8 deg).  What is STORED is two-level (ABI 19):
 *                                             phi_fix = sat32(P) in the 16-byte aircraft record — for -2^31 < P < 2^31 - 1, i.e.
 *                                               headings in (-76, 436) deg: the action space's [0, 360] and 76 deg either side,
 *                                               this IS the heading and nothing else is read or written;
 *                                             a record whose phi_fix is INT32_MIN / INT32_MAX is WIDE: its exact P is the
 *                                               integer-valued float64 atc_state_t.phi_wide[i][0] (written by the step that
 *                                               left the 32-bit range; float64 holds every integer up to 2^53).
 *                                           The last accepted heading target (last_action[2], atc_gym.py:311) is stored the same way:
 *                                           sat32 in last_act[i][1], exact value in phi_wide[i][1] when that is saturated.
 *   - a target is  counts = trunc(a * m + c):  ONE float64 fma on the fp32 action (m, c = the reference's factor / offset of
 *     atc_gym.py:64-78,318-335 in counts, both integers; exact for every fp32 action: 24 x 31 bits), truncated toward zero,
 *     NaN -> 0.  The speed's target is converted to uint32 SATURATING (gfx950: v_cvt_u32_f64): an action far outside the action
 *     space pins it at the end of the format's range, where it is refused like any target outside [100, 300] kt.  The heading's
 *     target is clamped to +-2^52 counts (|a_phi| <= 2.98e6 continuous, 5.4e8 discrete) — the one bound left: an aircraft-step
 *     whose heading target was clamped carries ATC_F_PHI_LIMIT; the aircraft still turns towards it at the rate limit, and all
 *     that differs from the reference is its action counter when two successive targets are BOTH beyond the bound and differ
 *     (tests/test_oracle_golden.py pins that step).
 *   - rate limits (model.py:75-78, 117-120) are exact integer arithmetic:  fix += clamp(target - fix, +-rint(rate dt 2^23))
 *     (wrapping 32-bit difference for the speed — speeds an Airplane can have, [100, 300] kt (model.py:22-23: the constructor
 *     raises outside), and the initial last_action 0 are < 2^31 counts apart; a speed placed from outside must lie inside
 *     [44, 356] kt, within 256 kt of every acceptable target: the host mirror's set_v refuses others —, exact for the heading: the 32-bit saturating
 *     form gives the same result whenever target and heading are both inside the 32-bit range, which is what the kernels
 *     evaluate for such lanes); the "action taken" discriminator (atc_gym.py:84,305-306) compares the same integer differences
 *     with 5 * 2^23 / 0.5 * 2^23 counts;
 *   - what the heading feeds: the kinematics and every relative angle / the corridor window are periodic in the heading, the
 *     observation's raw heading (atc_gym.py:269) is not.  A heading inside the 32-bit range is used as it is:
 *       fp32 heading = fmaf((float)phi_fix, 2^-23, 180),  kinematics from phi_fix (below).
 *     A WIDE heading P is first wrapped, in float64 like the kinematics' own reduction:  Pw = fma(rint(P * ATC_PHI_INV_TURN),
 *       -ATC_PHI_TURN, P)  (exact: the remainder of P modulo 360 deg in counts, |Pw| <= 180 deg within a count):
 *       kinematics, relative angles and the corridor window use Pw exactly like a phi_fix;  observation word 3 is
 *       (float)(180 + P 2^-23) evaluated in float64 (exact) and rounded once — the float32 of the reference's float64 heading.
 *   - ONE angle decides a flag: the window of Corridor._inside_corridor_angle (model.py:212-231), `min_angle <= relative_angle <= 45`
 *     with min_angle = arccos(dir_rwy . dir_plane) in radians — in exact arithmetic 0 <= rel <= 45 deg.  Its relative angle is evaluated
 *     EXACTLY, on the counts:  rel = wrap(P - Q)  (wrap as above, float64 on integers), Q = rint((phi_to_runway - 180) 2^23);  the window
 *     holds for 0 < rel <= 45 * 2^23 and, for rel == 0, iff ATC_C_ALIGNED_OK (the reference's own rounding luck for an exactly aligned
 *     heading, evaluated on the host with its expression).  An fp32 heading could not decide it: an aircraft told to fly the runway
 *     heading holds it to within the fp32 rounding of its ACTION (340 -/+ 1e-5 deg), one third of an fp32 ulp at 340 — the reference's
 *     float64 tells the two apart, and so must the flag (tests/golden/g11: the winning intercepts flown at heading - 360).  The
 *     relative angles of the observation and the shaping terms are values (1e-5 bar) and stay fp32;
 *   - the fp32 speed every other formula of the reference sees is (float)v_fix * 2^-23; both conversions are exact for every value
 *     with <= 24 significant bits, e.g. all integer speeds and headings;
 *   - the ALTITUDE (model.py:82-102) is the reference's float64, operation for operation (ABI 20; rounds 1-5 kept an fp32
 *     accumulator: exact for timesteps whose rate limits 41 dt / 15 dt are small multiples of an fp32 ulp — 1, 2, 5, 0.5 s — and
 *     0.4 ulp off PER STEP for 0.1 s: the below-MVA flag came one step late in 19 of 120 sustained descents, tests/golden/g12):
 *       target = a * m + c        one float64 fma on the fp32 action (the product is exact): the reference's a * f / 2 + f / 2 + off
 *                                 (atc_gym.py:333-335; discrete: a * 100 + 0, :329-330) bit for bit
 *       refused iff target < h_min or target > h_max                                        (model.py:91-94, float64 compares)
 *       h += max(min(target - h, 15 dt), -41 dt)       three float64 operations, dt the float64 of atc_params_t (model.py:95-102)
 *       below the MVA iff h < mva                       (atc_gym.py:149-153; the MVA height is an integer)
 *       observation words 2 and 5 = (float)h and (float)(h - mva), the difference in float64          (atc_gym.py:266,276)
 *     so every flag that depends on the altitude alone is the reference's for ANY timestep, including the ties it decides by its
 *     own accumulated rounding (15 000 ft - 3 000 x 4.1 ft against the 2 700 ft MVA at dt = 0.1).  What other formulas see — the
 *     glide-path test of the corridor (model.py:201-208), the shaping terms, the separation scan — is (float)h, one rounding.
 *     The last accepted altitude target (last_action[1], atc_gym.py:311) is kept as the same float64.
 *   - state placed from outside (entry points, fixtures) is the NEAREST count.
 * Heading kinematics (model.py:122-129, 345-348) in float64, shared bit for bit by every fp32 implementation (the HIP
 * kernels, the fp32 instantiation of the test oracle):
 *   k = rint(phi_fix * ATC_KIN_INV180)  [(180 * 2^23)^-1, nearest-even],  t = fma(k, -180 * 2^23, phi_fix)  (exact: the
 *   remainder in counts, |t| <= 90 * 2^23),  u = t * t,
 *   sin = t * (S0 + u (S1 + u (S2 + u (S3 + u (S4 + u S5))))),  cos = 1 + u (C1 + u (C2 + u (C3 + u (C4 + u C5))))
 *   (Horner, every step one fma; coefficients below: near-minimax in r = t pi / (180 * 2^23) on |r| <= pi/2, scaled to
 *   counts — tools/fit_kinematics_f64.py; max error 2.6e-11 / 4.4e-10);
 *   distance in position-grid counts  d = (double)v_fix * DA,  DA = ((double)dt / 3600) * 2^(k_pos - 23),  NEGATED when k is
 *   even (phi = 180 (1 + k) + t: an even k is an odd number of half turns);
 *   DITHERED ROUNDING:  x_fix += floor(sin * d + u11),  y_fix += floor(cos * d + u11)  (saturating adds), u11 = the low 11 bits of
 *   the env's time step (after its increment, atc_gym.py:135) bit-reversed, as a fraction in [0, 1) — the van der Corput
 *   sequence.  Plain rounding repeats the SAME round-off every step of a straight leg at constant speed (measured: up to
 *   7.8e-6 nm per episode); dithered, the round-offs of a constant displacement cancel to O(log n) counts over n steps.
 *   Evaluated as  r = fma(sin, d, M)  with M = the float64 whose high word is ATC_DITHER_MAGIC_HI (1.5 * 2^41) and whose low
 *   word is u11, i.e. M = 1.5 * 2^41 + u11 * 2^-11 (2^-11 is the last place of a float64 of that magnitude); the integer part of r is
 *   bits 11..42 of its pattern:  counts = (int32)(bits(r) >> 11).
 *   Requires 0.1423 dt 2^k_pos < 2^30 (512 kt: the displacement must fit 2^30 counts) and 5 dt < 256 (the speed's rate limit in
 *   counts): dt <= 51 s for any sector; atc_step refuses larger steps.
 * Measured against the float64 reference over the 650 963 steps of tests/golden/g9_wide.npz (tools/position_error.py):
 * positions within 1.5e-7 nm (median) / 1.04e-6 nm (maximum, 6 000-step episodes), speed and heading within one count;
 * the fixture replays at the plain 1e-5 bar everywhere — no near-FAF exception.
 *
 * Separation scan horizon (round 5; an implementation note, not a format: results are what a scan in every step gives).
 * Conflict = (d^2 < sep_nm^2) and (|dh| < sep_ft) for a pair of aircraft under control, d^2 = fma(dx, dx, dy dy) and dh from the
 * fp32 positions / altitudes.  The kinematics above bound what one step can do to a pair: an aircraft at or below 300 kt stays at or
 * below 300 kt (accepted speed targets lie in [100, 300] kt and the rate-limited move never overshoots) and moves by at most
 * (300 / 3600) dt nm (+ one position count per axis); its altitude moves by at most 15 dt ft up / 41 dt ft down (+ half an ulp).  A
 * pair with  d >= |sep_nm| + n (600 / 3600) dt  or  |dh| >= sep_ft + n 56 dt  (+ slack: 1e-5 relative + 1e-3 nm / 1 ft, hundreds of
 * times the fp32 evaluation error) therefore cannot be in conflict during the next n steps, whatever the actions.  Multi-step launches
 * (atc_rollout, atc_rollout_hold) of envs of more than 16 aircraft that do not report min_sep use this: a full scan notes which
 * groups of partners hold a pair inside those thresholds, and for the next n steps only those groups are scanned (none: no scan) —
 * unless an aircraft is above 300 kt or an altitude is beyond 2^17 ft in magnitude or NaN (then every step scans in full); an env
 * that is reset inside the launch ends the horizon of its wavefront.  tests/test_hip_edge_cases.py
 * (test_scan_horizon_on_the_fastest_closing_courses) drives pairs at exactly these closing rates through every phase of a horizon. */
#define ATC_V_FIX_SHIFT 23
#define ATC_PHI_FIX_SHIFT 23
#define ATC_PHI_FIX_OFFSET 180.0f
#define ATC_DITHER_MAGIC_HI 0x42880000u   /* high word of 1.5 * 2^41: the low word's low 11 bits are the dither fraction */
#define ATC_KIN_INV180 (0x1.6c16c16c16c17p-31)
#define ATC_KIN_HALF_TURN 1509949440.0    /* 180 * 2^23 */
#define ATC_PHI_TURN 3019898880.0         /* 360 * 2^23: a full turn in heading counts */
#define ATC_PHI_INV_TURN (0x1.6c16c16c16c17p-32)   /* RN(1 / ATC_PHI_TURN) */
#define ATC_PHI_LIMIT 4503599627370496.0  /* 2^52: heading targets are clamped to +-this (ATC_F_PHI_LIMIT) */
#define ATC_KIN_S0 (0x1.1df46a2514d5fp-29)
#define ATC_KIN_S1 (-0x1.dbb820c160971p-90)
#define ATC_KIN_S2 (0x1.dad945dddaa3dp-152)
#define ATC_KIN_S3 (-0x1.c366787a5df6bp-215)
#define ATC_KIN_S4 (0x1.f410ee5d4f8bep-279)
#define ATC_KIN_S5 (-0x1.5a91d219ad65ap-343)
#define ATC_KIN_C1 (-0x1.3f6a1d6c2409bp-59)
#define ATC_KIN_C2 (0x1.09b109e7441cbp-120)
#define ATC_KIN_C3 (-0x1.6198137ae3b3bp-183)
#define ATC_KIN_C4 (0x1.f762cae337122p-247)
#define ATC_KIN_C5 (-0x1.a6ebd88a524cfp-311)
#define ATC_POS_MAX_K 27

/* Persistent environment state (all device pointers).  Aircraft arrays are indexed env * N + k and packed so that a
 * wavefront moves each with one access per lane on consecutive addresses.  Per aircraft-step the step kernel reads
 * 16 + 8 + 12 B (record, altitude, action) and writes 16 + 8 B; the 16-byte last-action record is read and written only by
 * steps that may change it (not under ATC_M_ACTIONS_HELD, not inside a held block of a multi-step launch). */
typedef struct atc_state {
    int32_t* ac;      /* [B*N][4]  x_fix, y_fix (position grid counts, see above), phi_fix (heading, never wrapped, model.py:35-36;
                         fixed point, INT32_MIN / INT32_MAX = WIDE: see above and phi_wide), v_fix (speed, model.py:37; unsigned
                         counts) — one 16-byte record */
    double* alt;      /* [B*N]     altitude h [ft] (model.py:34): the reference's float64 (ABI 20, see "the ALTITUDE" above) */
    int32_t* last_act;/* [B*N][4]  last accepted v / phi / h targets = AtcGym.last_action (atc_gym.py:86,311) in the state's
                         own formats: word 0 v_fix, word 1 phi_fix (0 kt / 0 deg of atc_gym.py:86 = the counts of 0, not the
                         integer 0), words 2..3 the altitude target as a float64 — one 16-byte record */
    int32_t* env;     /* [B][ATC_ENV_WORDS]  per-step env record, see ATC_ENV_* */
    int32_t* stats;   /* [B][ATC_STAT_WORDS] per-episode env record, see ATC_STAT_* (touched only when an episode ends) */
    double* phi_wide; /* [B*N][4]  side record of aircraft whose 32-bit heading fields are saturated (see "Speed and heading": WIDE):
                         word 0 = exact heading counts, word 1 = exact last heading target (integer-valued float64, valid while
                         phi_fix / last_act[i][1] is INT32_MIN / INT32_MAX), word 2 = scratch of the step kernel (the wrapped
                         counts and observation word 3 of word 0, handed from the first half of a step to the second), word 3
                         reserved.  Never read or written for an aircraft whose heading and heading targets stay inside
                         (-76, 436) deg — every action inside the action space —, so it costs memory (32 B per aircraft), not
                         traffic; contents are unspecified while the 32-bit field is in range. */
} atc_state_t;
/* per-step env record (4 x 32-bit words; float fields are stored by bit pattern) */
enum {
    ATC_ENV_TIMESTEPS = 0,     /* i32  atc_gym.py:39 */
    ATC_ENV_ACTIONS_TAKEN = 1, /* i32  atc_gym.py:31 */
    ATC_ENV_TOTAL_REWARD = 2,  /* f32  atc_gym.py:30 */
    ATC_ENV_MASK_LO = 3,       /* u32  active mask bits 0..31: bit k = aircraft k still under control (extension) */
    ATC_ENV_WORDS = 4
};
/* per-episode env record (8 x 32-bit words) */
enum {
    ATC_STAT_EPISODES = 0,     /* i32  atc_gym.py:33 (_episodes_run) */
    ATC_STAT_EP_LENGTH = 1,    /* i32  length of the last finished episode (Monitor 'l') */
    ATC_STAT_EP_RETURN = 2,    /* f32  return of the last finished episode (Monitor 'r') */
    ATC_STAT_WIN_BITS = 3,     /* u32  last 10 episode outcomes, bit0 = most recent (atc_gym.py:36-37,359-363) */
    ATC_STAT_EP_ACTIONS = 4,   /* i32  actions_taken of the last finished episode: actions_per_timestep (atc_gym.py:197)
                                       keeps its last value across reset() = EP_ACTIONS / EP_LENGTH */
    ATC_STAT_MASK_HI = 5,      /* u32  active mask bits 32..63 (read / written per step only by envs of more than 32 aircraft) */
    ATC_STAT_WORDS = 8
};

/* Per-step outputs (device pointers; nullable ones may be NULL). */
typedef struct atc_out {
    float* obs;        /* [B*N*10] what step() returns (normalised iff ATC_M_NORMALIZE), atc_gym.py:187-192 */
    float* raw_obs;    /* nullable [B*N*10] info["original_state"], atc_gym.py:192 */
    float* reward;     /* [B] env reward = sum over aircraft */
    float* ac_reward;  /* nullable [B*N] per-aircraft reward */
    uint8_t* done;     /* [B] */
    uint16_t* flags;   /* [B*N] ATC_F_* */
    float* min_sep;    /* nullable [B] minimum horizontal separation among active pairs [nm] (diagnostic) */
    float* term_obs;   /* nullable [B*N*10] terminal observation of envs that were auto-reset this step */
    uint32_t* packet;  /* nullable, N == 1 only: [B][ATC_PKT_CHUNKS][4] — the step result of one env as self-validating
                          16-byte chunks, for a host that polls pinned mapped memory instead of synchronising the stream
                          (the single-env AtcGym).  Every chunk = 3 payload words + params.reserved0 (the caller's step
                          sequence number) and is written with ONE 16-byte store, so a reader that sees the expected tag in
                          a chunk and THEN reads its payload has that step's values whatever order the chunks arrive in:
                          obs[10], raw obs[10], reward | flags + (done << 16), timesteps, actions_taken | x, y grid counts */
} atc_out_t;
#define ATC_PKT_CHUNKS 9

/* Opaque device-resident scenario (owned by the library). */
typedef struct atc_scenario atc_scenario_t;

/* -- lifecycle -------------------------------------------------------------------------------- */
int atc_abi_version(void);
const char* atc_last_error(void);

/* Uploads a compiled scenario blob (host pointer, n_words floats) to `device`.
 * Replaces: AtcGym.__init__ scenario unpacking, atc_gym.py:45-58. */
int atc_scenario_create(const float* blob_host, size_t n_words, int device, atc_scenario_t** out);
int atc_scenario_destroy(atc_scenario_t* s);
/* Attaches (copies to the device) an LDS-resident lookup table — see "LDS-resident lookup table" above — or detaches the current
 * one (table_host == NULL).  Every code and offset is validated here; -1 if the table is malformed, larger than the device's LDS
 * per workgroup, or the scenario has no lookup grid / has noise-abatement areas.  With a table attached, atc_rollout[_hold] of
 * one-aircraft envs with at most 256 envs per CU of the device (65 536 on MI355X) stage it in LDS; every other launch ignores it.
 * Not thread-safe against launches that use the same handle concurrently (attach before stepping). */
int atc_scenario_attach_lds_table(atc_scenario_t* s, const void* table_host, size_t n_bytes);

/* Zero-copy for latency-bound callers (the single-env AtcGym, atc_gym.py:128-192, whose step is one aircraft): every
 * state / action / output pointer may also address pinned host memory (hipHostMalloc, e.g. a torch tensor with
 * pin_memory()) that is mapped into the device's address space; the kernels then read and write it over the host link
 * and no copy is launched around the step.  *dev receives the device address of such a buffer; fails (-2) if `host`
 * is not mapped pinned memory. */
int atc_host_mapped_ptr(const void* host, void** dev);

/* -- batched queries (same device functions as the step kernel) -------------------------------- */
/* Airspace.get_mva_height, model.py:282-292.  out_h[i] = MVA height [ft] or -1 (outside airspace).
 * use_grid != 0 uses the lookup grid if the blob has one. */
int atc_query_mva(const atc_scenario_t* s, int n, const float* x, const float* y, int32_t* out_h, int use_grid,
                  void* stream);
/* Airspace.find_mva, model.py:282-289: out_idx[i] = index of the MVA polygon in list order, or -1. */
int atc_query_mva_index(const atc_scenario_t* s, int n, const float* x, const float* y, int32_t* out_idx, int use_grid,
                        void* stream);
/* Airspace.get_mva_height through the attached LDS table (diagnostic / test entry): out_h as atc_query_mva; from_lds[i] (may be
 * NULL) = 1 where the table answered, 0 where the point's wavefront (64 consecutive points) went to the lookup grid. */
int atc_query_mva_lds(const atc_scenario_t* s, int n, const float* x, const float* y, int32_t* out_h, uint8_t* from_lds,
                      void* stream);
/* Runway.inside_corridor, model.py:248-257,188-231.  angle_only != 0 evaluates Corridor._inside_corridor_angle
 * (model.py:212-231) alone.  out[i] = 0/1. */
int atc_query_corridor(const atc_scenario_t* s, int n, const float* x, const float* y, const float* h,
                       const float* phi, int angle_only, uint8_t* out, void* stream);
/* Shaping rewards, atc_gym.py:199-260: out3[3*i+0..2] = (_reward_approach_position, _reward_approach_angle,
 * _reward_glideslope) for inputs (d_faf, phi_rel_faf, phi_plane, h, on_gp_altitude). */
int atc_query_shaping(const atc_scenario_t* s, int n, const float* d_faf, const float* phi_rel_faf,
                      const float* phi_plane, const float* h, const float* on_gp, float* out3, void* stream);

/* -- environment ------------------------------------------------------------------------------- */
/* AtcGym.reset, atc_gym.py:337-365, for every env whose mask byte is non-zero (mask == NULL: all envs).
 * Writes the RAW reset observation (mva = 0, quirk atc_gym.py:351,365) to `obs` for reset envs only.
 * first != 0 additionally clears the last-action fields / win_bits / episodes (what AtcGym.__init__ does,
 * atc_gym.py:29-41,86). */
int atc_reset(const atc_scenario_t* s, int B, int N, const atc_state_t* st, const uint8_t* mask, float* obs,
              const atc_params_t* p, int first, void* stream);

/* AtcGym._get_state(0), atc_gym.py:262-277,351: RAW observation (mva = 0) of the CURRENT aircraft state of every masked
 * env (mask == NULL: all) — used after the host places aircraft explicitly (e.g. to replay the reference's
 * random.choice entry draws, atc_gym.py:346-348). */
int atc_observe(const atc_scenario_t* s, int B, int N, const atc_state_t* st, const uint8_t* mask, float* obs,
                const atc_params_t* p, void* stream);

/* AtcGym.step, atc_gym.py:128-192, for B envs x N aircraft.  actions: [B*N*3] (v,h,phi per aircraft),
 * continuous in [-1,1] or discrete indices stored as floats (atc_gym.py:318-335). */
int atc_step(const atc_scenario_t* s, int B, int N, const atc_state_t* st, const float* actions,
             const atc_out_t* out, const atc_params_t* p, void* stream);

/* One step of several INDEPENDENT sub-batches with a single call — the arguments of n atc_step calls, each with its own
 * state, outputs and (normally) its own stream.  Meant for pipelined callers that keep a few sub-batches in flight on
 * separate streams with no join between steps: the launch ramp and tail of one sub-batch then overlap the body of the
 * others (65 536 x 16 as 2-4 sub-batches: 21.6-23.4 us per step of all envs instead of 24.9 us), and the host pays one
 * foreign call per step instead of n.  Stops at the first failing launch and returns its error. */
typedef struct atc_step_call {
    const atc_scenario_t* s;
    int32_t B, N;
    const atc_state_t* st;
    const float* actions;
    const atc_out_t* out;
    const atc_params_t* p;
    void* stream;
} atc_step_call_t;
int atc_step_multi(int n, const atc_step_call_t* calls);

/* atc_step of ONE env x ONE aircraft (B = N = 1) whose outputs include out->packet in pinned mapped memory, followed by the
 * wait for its result in the same foreign call.  `actions` is a HOST pointer to the 3 action values here: they are read
 * during the call and travel with the kernel arguments (no read over the host link on the device side).
 * The step is launched with p->reserved0 = seq, then `packet_host` (the HOST
 * address of the same packet buffer) is polled until all ATC_PKT_CHUNKS chunks carry the tag `seq`, and their 27 payload
 * words are copied to payload[27] (see atc_out_t.packet for their meaning).  Returns 0, the launch's error, or -3 when the
 * result has not arrived within `timeout_us` microseconds (the caller then synchronises the stream and reads the packet itself).
 * `seq` must differ from the previous call's; `packet_host` must be 16-byte aligned (each chunk is read with one 16-byte load, so
 * a tag and the payload words it validates always come from the same access).  The kernel's trailing state stores may still be in flight on return: anything
 * else that touches the env's buffers has to drain the stream first.  Replaces the wait in AtcGym.step (atc_gym.py:128-192). */
int atc_step_packet(const atc_scenario_t* s, const atc_state_t* st, const float* actions, const atc_out_t* out,
                    atc_params_t* p, uint32_t seq, const uint32_t* packet_host, uint32_t* payload, int timeout_us, void* stream);

/* Persistent step server for ONE env x ONE aircraft (ABI 20) — what the drop-in AtcGym steps through (atc_gym.py:128-192; the
 * reference's callers step one env per process: learning/atc-gym-stable-baselines.py:69-80, atc-gym-compute-performance.py:10-19).
 * atc_serve_start launches ONE resident wavefront that keeps the env's state in registers and polls `mailbox` — 64 bytes of pinned
 * mapped host memory (hipHostMalloc / a pinned torch tensor), 64-byte aligned — for commands; atc_serve_step writes {3 action
 * floats, seq} into it with one 16-byte store and polls out->packet (see atc_out_t.packet; `packet_host` / `payload` as in
 * atc_step_packet) for the 9 chunks tagged `seq`; no kernel launch, argument upload or state round trip per step.
 *   mailbox words: 0..2 action (float bits), 3 sequence number (0xffffffff = quit) — written by the host;
 *                  4 server state (0 not yet running, 1 serving, 2 left: lease ran out, 3 left: quit), 5 last served sequence
 *                  number — written by the device.
 *   - `seq` of atc_serve_start = the sequence number of the LAST step already taken (the first atc_serve_step uses seq + 1, every
 *     further one the previous + 1); 0xffffffff is reserved.
 *   - the server leaves by itself after `lease_us` microseconds (>= 10) without a command: atc_serve_step then returns -4 for the
 *     command it did not see and the caller starts the server again (same seq rule) and repeats the call.  -3: no answer within
 *     `timeout_us` (the caller falls back to atc_serve_stop + atc_step_packet).
 *   - CHOOSING THE LEASE.  A resident kernel occupies a hardware queue; a process has few (4 by default) and HIP streams beyond that
 *     share them: work submitted to a stream that shares the server's queue waits until the server has left.  A caller that steps
 *     in a tight loop and submits nothing else (the reference's FPS script) can use any lease; a caller that interleaves other GPU
 *     work keeps it at the scale of a kernel launch — envs.atc.atc_gym.AtcGym uses 50 us and only serves while its steps follow
 *     each other within 50 us (otherwise it steps by atc_step_packet): the worst a colliding stream can wait is one lease.
 *   - WHILE THE SERVER RUNS the env's state lives in its registers: nothing else may read or write the env's atc_state_t buffers or
 *     launch on `stream` (a launch would queue behind the resident kernel).  atc_serve_stop sends quit and synchronises the stream:
 *     after it the state is in memory as after an atc_step.  Outputs other than the packet (obs, reward, ...) are written every step
 *     like atc_step's.
 * Results are those of atc_step on the same state and actions, bit for bit (same device functions). */
int atc_serve_start(const atc_scenario_t* s, const atc_state_t* st, const atc_out_t* out, const atc_params_t* p,
                    uint32_t* mailbox_host, uint32_t seq, int lease_us, void* stream);
int atc_serve_step(uint32_t* mailbox_host, const float* actions, uint32_t seq, const uint32_t* packet_host, uint32_t* payload,
                   int timeout_us);
int atc_serve_stop(uint32_t* mailbox_host, void* stream);

/* T consecutive steps in ONE launch with aircraft state held in registers.  actions: [T][B*N*3];
 * outputs are [T][...] versions of atc_out_t (each pointer strides by its per-step size).
 * Requires ATC_M_AUTO_RESET semantics to be meaningful for T > episode length. */
int atc_rollout(const atc_scenario_t* s, int B, int N, int T, const atc_state_t* st, const float* actions,
                const atc_out_t* out, const atc_params_t* p, void* stream);

/* The same with every action HELD for `hold` consecutive steps (frame skip — the protocol of the reference's demo loop,
 * learning/atc-gym-demo.py:18-19: one sampled action is applied 20 times): actions: [T / hold][B*N*3], step t uses
 * block t / hold; T must be a multiple of hold (ATC_ERR_ARG otherwise: a partial last block is refused, not read).
 * Results are identical to atc_rollout with each block repeated `hold` times; the action tensor and its
 * HBM traffic shrink by that factor. */
int atc_rollout_hold(const atc_scenario_t* s, int B, int N, int T, int hold, const atc_state_t* st, const float* actions,
                     const atc_out_t* out, const atc_params_t* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ATC_STEP_H */
