/* hanabi_batch.h -- C ABI of libhanabi_batch.so: a batched Hanabi stepper.
 *
 * Host code (no device work): N independent Hanabi tables advance in ONE call and write their observation,
 * centralised-observation and legal-action rows straight into caller-owned [N, dim] float32 arrays -- the layout the
 * rollout buffer's insert() takes -- instead of N Python env objects behind N pipes, each building Python lists.
 *
 * What it replaces in the reference (marlbenchmark/on-policy):
 *   - the game rules, chance (card dealing) and the canonical observation encoder of the vendored
 *     hanabi_learning_environment: onpolicy/envs/hanabi/hanabi_lib/hanabi_state.cc:84-96 (DealCard), :164-283 (MoveIsLegal,
 *     ApplyMove), :364-382 (Score, EndOfGameStatus); hanabi_game.cc:108-114 (PickRandomChance), :162-196 (move uids);
 *     canonical_encoders.cc:64-115 (hands), :133-182 (board), :204-230 (discards), :259-342 (last action),
 *     :377-433 + :482-536 (card knowledge / "V0 belief"), :575-594 (own hand);
 *   - its C binding onpolicy/envs/hanabi/pyhanabi.h:26-197 (one heap object per move / observation / encoding string);
 *   - the per-env Python glue onpolicy/envs/hanabi/Hanabi_Env.py:278-312 (reset), :451-500 (step).
 * Results are identical to the reference engine's for the same seeds and actions (tests/golden/hanabi_cases.npz):
 * the deal uses the same std::mt19937 / std::discrete_distribution draws, and the encoder reproduces the reference's
 * integer arithmetic in the belief section (plausible-card bit x remaining count, divided by the row total and
 * TRUNCATED to int, canonical_encoders.cc:509-530 -- i.e. 1 only where a single card type remains possible).
 *
 * All functions return 0 on success or a negative hanabi_batch_error code; nothing throws across the boundary.
 * A batch is not thread-safe; distinct batches are independent.
 */
#ifndef HANABI_BATCH_H_
#define HANABI_BATCH_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hanabi_batch hanabi_batch_t;

/* observation_type: hanabi_game.h:40 (0 minimal: no belief section; 1 card knowledge; 2 seer) */
typedef struct hanabi_rules {
  int32_t colors;                 /* 1..5 */
  int32_t ranks;                  /* 1..5 */
  int32_t players;                /* 2..5 */
  int32_t hand_size;              /* 1..5; <= 0: 5 below four players, else 4 (hanabi_game.cc:155-160) */
  int32_t max_information_tokens; /* >= 0 */
  int32_t max_life_tokens;        /* >= 1 */
  int32_t observation_type;
  int32_t random_start_player;    /* 0: player 0 starts every game */
} hanabi_rules_t;

enum hanabi_batch_error {
  HANABI_OK = 0,
  HANABI_BAD_ARGUMENT = -1,
  HANABI_ILLEGAL_MOVE = -2,       /* the table named by hanabi_batch_failed_table() was left untouched */
  HANABI_NOT_STARTED = -3,        /* step / encode on a table that was never reset */
};

/* share_mode of hanabi_batch_encode (Hanabi_Env.py:300-305): */
enum hanabi_share_mode {
  HANABI_SHARE_OWN_HAND = 0,      /* [own hand | observation | turn]          (use_obs_instead_of_state = False) */
  HANABI_SHARE_ALL_PLAYERS = 1,   /* [observation of player 0 | 1 | ... | turn] (use_obs_instead_of_state = True)  */
};

/* seeds[n_tables]: one generator per table, seeded like HanabiGame (hanabi_game.cc:50); it keeps running across the
 * episodes of that table, as the reference's per-env game object does. */
hanabi_batch_t *hanabi_batch_create(const hanabi_rules_t *rules, int32_t n_tables, const int32_t *seeds);
void hanabi_batch_destroy(hanabi_batch_t *batch);

int32_t hanabi_batch_tables(const hanabi_batch_t *batch);
int32_t hanabi_batch_players(const hanabi_batch_t *batch);
int32_t hanabi_batch_num_moves(const hanabi_batch_t *batch);      /* HanabiGame::MaxMoves */
int32_t hanabi_batch_obs_len(const hanabi_batch_t *batch);        /* CanonicalObservationEncoder::Shape */
int32_t hanabi_batch_own_hand_len(const hanabi_batch_t *batch);   /* CanonicalObservationEncoder::OwnHandShape */
int32_t hanabi_batch_failed_table(const hanabi_batch_t *batch);   /* table index of the last HANABI_ILLEGAL_MOVE */

/* New game on every table with choose[i] != 0 (choose == NULL: all), cards dealt until a player is to move. */
int hanabi_batch_reset(hanabi_batch_t *batch, const uint8_t *choose);

/* actions[i]: move uid, or -1 to leave table i alone.  Writes, per table, the score differential (0 for idle
 * tables), status (0 running, 1 finished, 2 idle -- the reference env's done = None) and the current score.  All
 * moves are validated first; on an illegal one nothing is applied to any table. */
int hanabi_batch_step(hanabi_batch_t *batch, const int32_t *actions, float *rewards, uint8_t *status,
                      int32_t *scores);

/* Rows for the player to move of every table with active[i] != 0 (NULL: all); rows of the others are zero-filled:
 *   obs        [n, obs_len + players]                      observation | one-hot turn
 *   share_obs  [n, share_len + players]                    see hanabi_share_mode; share_len = own_hand_len + obs_len
 *                                                          or players * obs_len
 *   available  [n, num_moves]                              1 where the move uid is legal
 *   to_move    [n] (may be NULL)                           player index, -1 for inactive rows */
int hanabi_batch_encode(const hanabi_batch_t *batch, int share_mode, const uint8_t *active, float *obs,
                        float *share_obs, float *available, int32_t *to_move);

/* One table, one observer: the encoder output exactly as the reference returns it (ints), for tests and tools. */
int hanabi_batch_player_view(const hanabi_batch_t *batch, int32_t table, int32_t player, int32_t *obs,
                             int32_t *own_hand);

/* Table state for logging / tests: out[0..7] = life tokens, information tokens, deck size, score, player to move,
 * end-of-game status (hanabi_state.h: 0 not finished, 1 out of life tokens, 2 out of cards, 3 completed),
 * turns left once the deck is empty, number of discarded cards; out[8..8+colors) = fireworks. */
int hanabi_batch_table_state(const hanabi_batch_t *batch, int32_t table, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif  /* HANABI_BATCH_H_ */
