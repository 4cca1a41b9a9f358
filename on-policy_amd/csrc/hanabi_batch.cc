// Batched Hanabi stepper: N tables advanced per call, observations written as rows of caller-owned float32 arrays.
// C ABI and the reference file:line of every rule / encoder section: include/hanabi_batch.h.
//
// A table is a fixed-size POD (cards are indices color * ranks + rank, hint knowledge is two bitmasks per hand slot),
// so a batch is one contiguous vector and stepping or encoding it touches no heap.  The only pieces that have to be
// library-identical to the reference engine are the random draws: std::mt19937 + std::discrete_distribution over the
// remaining card counts (hanabi_game.cc:108-114), which is why this is C++ <random> and not a hand-rolled generator.
#include "../../include/hanabi_batch.h"

#include <algorithm>
#include <cstring>
#include <new>
#include <random>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxColors = 5, kMaxRanks = 5, kMaxPlayers = 5, kMaxHand = 5;
constexpr int kMaxCardTypes = kMaxColors * kMaxRanks;

enum MoveKind : int8_t { kNone = 0, kPlay, kDiscard, kHintColor, kHintRank };
enum EndOfGame { kRunning = 0, kOutOfLives, kOutOfCards, kCompleted };

struct Move {
  MoveKind kind;
  int8_t slot;     // play / discard: position in the hand
  int8_t offset;   // hints: target player relative to the mover (1 .. players - 1)
  int8_t value;    // hints: color or rank
};

struct Layout {    // everything derived from the rules once
  hanabi_rules_t r;
  int card_types, deck_total, num_moves;
  int copies[kMaxRanks];
  int hands_len, board_len, discard_len, last_len, belief_len, obs_len, own_len;
  int deck_bits;
};

struct Slot {
  uint8_t card;          // color * ranks + rank
  uint8_t colors_left;   // bit c set: no hint has ruled color c out
  uint8_t ranks_left;
  int8_t color_hint;     // -1 until a hint named this card's color
  int8_t rank_hint;
};

struct LastMove {        // the most recent player move, as the encoder's last-action section needs it
  MoveKind kind;
  int8_t player, slot, offset, value, card;
  uint8_t touched;       // hints: bit i = slot i matched
  bool scored, regained_token;
};

struct Table {
  std::mt19937 rng;
  uint8_t stock[kMaxCardTypes];
  int stock_total;
  Slot hand[kMaxPlayers][kMaxHand];
  uint8_t hand_n[kMaxPlayers];
  uint8_t fireworks[kMaxColors];
  uint8_t discarded[kMaxCardTypes];
  int discards;
  int info, lives;
  int to_move;           // -1 while a card is owed to some hand (chance node)
  int next_player;
  int turns_left;
  LastMove last;
  bool started;
};

}  // namespace

struct hanabi_batch {
  Layout L;
  std::vector<Table> tables;
  int failed_table;
};

namespace {

bool make_layout(const hanabi_rules_t &in, Layout *L) {
  hanabi_rules_t r = in;
  if (r.players < 2 || r.players > kMaxPlayers || r.colors < 1 || r.colors > kMaxColors || r.ranks < 1 ||
      r.ranks > kMaxRanks || r.max_information_tokens < 0 || r.max_life_tokens < 1 || r.observation_type < 0 ||
      r.observation_type > 2)
    return false;
  if (r.hand_size <= 0) r.hand_size = r.players < 4 ? 5 : 4;
  if (r.hand_size > kMaxHand) return false;
  L->r = r;
  L->card_types = r.colors * r.ranks;
  int per_color = 0;
  for (int k = 0; k < r.ranks; ++k) {   // hanabi_game.cc:139-148: three of the lowest rank, one of the highest, else two
    L->copies[k] = k == 0 ? 3 : (k == r.ranks - 1 ? 1 : 2);
    per_color += L->copies[k];
  }
  L->deck_total = per_color * r.colors;
  if (r.hand_size * r.players > L->deck_total) return false;
  L->num_moves = 2 * r.hand_size + (r.players - 1) * (r.colors + r.ranks);
  L->deck_bits = L->deck_total - r.players * r.hand_size;
  L->hands_len = (r.players - 1) * r.hand_size * L->card_types + r.players;
  L->board_len = L->deck_bits + L->card_types + r.max_information_tokens + r.max_life_tokens;
  L->discard_len = L->deck_total;
  L->last_len = r.players + 4 + r.players + r.colors + r.ranks + 2 * r.hand_size + L->card_types + 2;
  L->belief_len = r.observation_type == 0 ? 0 : r.players * r.hand_size * (L->card_types + r.colors + r.ranks);
  L->obs_len = L->hands_len + L->board_len + L->discard_len + L->last_len + L->belief_len;
  L->own_len = r.hand_size * L->card_types;
  return true;
}

inline Move move_of(const Layout &L, int uid) {
  const int h = L.r.hand_size;
  if (uid < 0 || uid >= L.num_moves) return {kNone, -1, -1, -1};
  if (uid < h) return {kDiscard, (int8_t)uid, -1, -1};
  uid -= h;
  if (uid < h) return {kPlay, (int8_t)uid, -1, -1};
  uid -= h;
  const int color_hints = (L.r.players - 1) * L.r.colors;
  if (uid < color_hints) return {kHintColor, -1, (int8_t)(1 + uid / L.r.colors), (int8_t)(uid % L.r.colors)};
  uid -= color_hints;
  return {kHintRank, -1, (int8_t)(1 + uid / L.r.ranks), (int8_t)(uid % L.r.ranks)};
}

inline int score_of(const Layout &L, const Table &t) {
  if (t.lives <= 0) return 0;
  int s = 0;
  for (int c = 0; c < L.r.colors; ++c) s += t.fireworks[c];
  return s;
}

inline EndOfGame end_of(const Layout &L, const Table &t) {
  if (t.lives < 1) return kOutOfLives;
  if (score_of(L, t) >= L.card_types) return kCompleted;
  if (t.turns_left <= 0) return kOutOfCards;
  return kRunning;
}

inline int hand_owed_a_card(const Layout &L, const Table &t) {
  for (int p = 0; p < L.r.players; ++p)
    if (t.hand_n[p] < L.r.hand_size) return p;
  return -1;
}

inline void pass_turn(const Layout &L, Table *t) {
  if (t->stock_total > 0 && hand_owed_a_card(L, *t) >= 0) {
    t->to_move = -1;
  } else {
    t->to_move = t->next_player;
    t->next_player = (t->to_move + 1) % L.r.players;
  }
}

bool legal(const Layout &L, const Table &t, const Move &m) {
  if (t.to_move < 0) return false;
  switch (m.kind) {
    case kDiscard:
      return t.info < L.r.max_information_tokens && m.slot < t.hand_n[t.to_move];
    case kPlay:
      return m.slot < t.hand_n[t.to_move];
    case kHintColor:
    case kHintRank: {
      if (t.info <= 0 || m.offset < 1 || m.offset >= L.r.players) return false;
      const int target = (t.to_move + m.offset) % L.r.players;
      for (int i = 0; i < t.hand_n[target]; ++i) {
        const int card = t.hand[target][i].card;
        if ((m.kind == kHintColor ? card / L.r.ranks : card % L.r.ranks) == m.value) return true;
      }
      return false;
    }
    default:
      return false;
  }
}

// One card from the stock to the first short hand.  The draw is the reference's: a discrete distribution over the
// card types still in stock, weights count / stock size, in card-index order (hanabi_state.cc:327-339 +
// hanabi_game.cc:108-114).  (With one type left the distribution consumes no random number -- same class, same rule.)
void deal_one(const Layout &L, Table *t) {
  double weight[kMaxCardTypes];
  int type_of[kMaxCardTypes];
  int n = 0;
  for (int i = 0; i < L.card_types; ++i) {
    if (t->stock[i] > 0) {
      weight[n] = static_cast<double>(t->stock[i]) / static_cast<double>(t->stock_total);
      type_of[n++] = i;
    }
  }
  std::discrete_distribution<std::mt19937::result_type> pick(weight, weight + n);
  const int card = type_of[pick(t->rng)];
  --t->stock[card];
  --t->stock_total;
  const int p = hand_owed_a_card(L, *t);
  Slot &s = t->hand[p][t->hand_n[p]++];
  s.card = (uint8_t)card;
  s.colors_left = (uint8_t)((1u << L.r.colors) - 1);
  s.ranks_left = (uint8_t)((1u << L.r.ranks) - 1);
  s.color_hint = s.rank_hint = -1;
  if (L.r.observation_type == 2) {      // seer: dealt cards are known outright (hanabi_state.cc:230-233)
    s.color_hint = (int8_t)(card / L.r.ranks);
    s.rank_hint = (int8_t)(card % L.r.ranks);
    s.colors_left = (uint8_t)(1u << s.color_hint);
    s.ranks_left = (uint8_t)(1u << s.rank_hint);
  }
  pass_turn(L, t);
}

void new_game(const Layout &L, Table *t) {
  for (int i = 0; i < L.card_types; ++i) t->stock[i] = (uint8_t)L.copies[i % L.r.ranks];
  t->stock_total = L.deck_total;
  std::memset(t->hand_n, 0, sizeof(t->hand_n));
  std::memset(t->fireworks, 0, sizeof(t->fireworks));
  std::memset(t->discarded, 0, sizeof(t->discarded));
  t->discards = 0;
  t->info = L.r.max_information_tokens;
  t->lives = L.r.max_life_tokens;
  t->to_move = -1;
  t->next_player = 0;
  if (L.r.random_start_player) {        // hanabi_game.cc:150-157
    std::uniform_int_distribution<std::mt19937::result_type> first(0, L.r.players - 1);
    t->next_player = (int)first(t->rng);
  }
  t->turns_left = L.r.players;
  t->last = LastMove{kNone, -1, -1, -1, -1, -1, 0, false, false};
  t->started = true;
  while (t->to_move < 0) deal_one(L, t);
}

inline void remove_slot(Table *t, int p, int slot) {
  for (int i = slot; i + 1 < t->hand_n[p]; ++i) t->hand[p][i] = t->hand[p][i + 1];
  --t->hand_n[p];
}

void apply(const Layout &L, Table *t, const Move &m) {
  if (t->stock_total == 0) --t->turns_left;          // hanabi_state.cc:220-222
  const int p = t->to_move;
  LastMove last{m.kind, (int8_t)p, m.slot, m.offset, m.value, -1, 0, false, false};
  switch (m.kind) {
    case kDiscard: {
      if (t->info < L.r.max_information_tokens) {
        ++t->info;
        last.regained_token = true;
      }
      const int card = t->hand[p][m.slot].card;
      last.card = (int8_t)card;
      ++t->discarded[card];
      ++t->discards;
      remove_slot(t, p, m.slot);
      break;
    }
    case kPlay: {
      const int card = t->hand[p][m.slot].card;
      const int color = card / L.r.ranks, rank = card % L.r.ranks;
      last.card = (int8_t)card;
      if (rank == t->fireworks[color]) {
        last.scored = true;
        if (++t->fireworks[color] == L.r.ranks && t->info < L.r.max_information_tokens) {
          ++t->info;                                   // finished stack hands a token back
          last.regained_token = true;
        }
      } else {
        --t->lives;
        ++t->discarded[card];
        ++t->discards;
      }
      remove_slot(t, p, m.slot);
      break;
    }
    case kHintColor:
    case kHintRank: {
      --t->info;
      const int target = (p + m.offset) % L.r.players;
      for (int i = 0; i < t->hand_n[target]; ++i) {
        Slot &s = t->hand[target][i];
        if (m.kind == kHintColor) {
          if (s.card / L.r.ranks == m.value) {
            last.touched |= (uint8_t)(1u << i);
            s.color_hint = m.value;
            s.colors_left = (uint8_t)(1u << m.value);
          } else {
            s.colors_left &= (uint8_t)~(1u << m.value);
          }
        } else {
          if (s.card % L.r.ranks == m.value) {
            last.touched |= (uint8_t)(1u << i);
            s.rank_hint = m.value;
            s.ranks_left = (uint8_t)(1u << m.value);
          } else {
            s.ranks_left &= (uint8_t)~(1u << m.value);
          }
        }
      }
      break;
    }
    default:
      break;
  }
  t->last = last;
  pass_turn(L, t);
  while (t->to_move < 0) deal_one(L, t);
}

// ---------------------------------------------------------------------------------------------- encoder
// Writes through a functor so that the same code fills float rows (batched path) and int vectors (player_view).
template <typename T>
void encode_view(const Layout &L, const Table &t, int observer, T *out) {
  const hanabi_rules_t &r = L.r;
  const int P = r.players, H = r.hand_size, K = L.card_types;
  std::fill(out, out + L.obs_len, T(0));
  T *o = out;
  // other players' cards, observer-relative order; then one "hand is short" bit per player
  for (int off = 1; off < P; ++off) {
    const int p = (observer + off) % P;
    for (int i = 0; i < t.hand_n[p]; ++i) o[i * K + t.hand[p][i].card] = T(1);
    o += H * K;
  }
  for (int off = 0; off < P; ++off)
    if (t.hand_n[(observer + off) % P] < H) o[off] = T(1);
  o += P;
  // board: deck thermometer, fireworks one-hot (highest rank played), token thermometers
  for (int i = 0; i < t.stock_total; ++i) o[i] = T(1);
  o += L.deck_bits;
  for (int c = 0; c < r.colors; ++c) {
    if (t.fireworks[c] > 0) o[t.fireworks[c] - 1] = T(1);
    o += r.ranks;
  }
  for (int i = 0; i < t.info; ++i) o[i] = T(1);
  o += r.max_information_tokens;
  for (int i = 0; i < t.lives; ++i) o[i] = T(1);
  o += r.max_life_tokens;
  // discards: thermometer per card type, as many bits as the type has copies
  for (int k = 0; k < K; ++k) {
    for (int i = 0; i < t.discarded[k]; ++i) o[i] = T(1);
    o += L.copies[k % r.ranks];
  }
  // last player move
  if (t.last.kind != kNone) {
    const LastMove &m = t.last;
    const int who = (m.player - observer + P) % P;
    T *q = o;
    q[who] = T(1);
    q += P;
    q[m.kind == kPlay ? 0 : m.kind == kDiscard ? 1 : m.kind == kHintColor ? 2 : 3] = T(1);
    q += 4;
    const bool hint = m.kind == kHintColor || m.kind == kHintRank;
    if (hint) q[(who + m.offset) % P] = T(1);
    q += P;
    if (m.kind == kHintColor) q[m.value] = T(1);
    q += r.colors;
    if (m.kind == kHintRank) q[m.value] = T(1);
    q += r.ranks;
    if (hint)
      for (int i = 0; i < H; ++i)
        if (m.touched & (1u << i)) q[i] = T(1);
    q += H;
    if (!hint) q[m.slot] = T(1);
    q += H;
    if (!hint) q[m.card] = T(1);
    q += K;
    if (m.kind == kPlay) {
      if (m.scored) q[0] = T(1);
      if (m.regained_token) q[1] = T(1);
    }
  }
  o += L.last_len;
  // card knowledge of every hand slot (observer first).  The reference weights the possible-card bits by how many
  // copies are neither discarded nor on the fireworks, then divides by the row total IN INTEGERS
  // (canonical_encoders.cc:509-530): what survives is a 1 where exactly one card type is still possible, 0 elsewhere.
  if (L.belief_len) {
    int remaining[kMaxCardTypes], color_of[kMaxCardTypes], rank_of[kMaxCardTypes];
    for (int k = 0; k < K; ++k) {
      const int color = color_of[k] = k / r.ranks, rank = rank_of[k] = k % r.ranks;
      remaining[k] = L.copies[rank] - t.discarded[k] - (rank < t.fireworks[color] ? 1 : 0);
    }
    const int per_slot = K + r.colors + r.ranks;
    for (int off = 0; off < P; ++off) {
      const int p = (observer + off) % P;
      for (int i = 0; i < t.hand_n[p]; ++i) {
        const Slot &s = t.hand[p][i];
        T *q = o + i * per_slot;
        // (int)(w_k / sum w) is 1 exactly when w_k is the only non-zero weight (all values are small integers,
        // exact in float), so the division is a count of the card types that are possible AND still unaccounted for
        int candidates = 0, only = -1;
        for (int k = 0; k < K; ++k) {
          if (remaining[k] > 0 && ((s.colors_left >> color_of[k]) & 1) && ((s.ranks_left >> rank_of[k]) & 1)) {
            ++candidates;
            only = k;
          }
        }
        if (candidates == 1) q[only] = T(1);
        if (s.color_hint >= 0) q[K + s.color_hint] = T(1);
        if (s.rank_hint >= 0) q[K + r.colors + s.rank_hint] = T(1);
      }
      o += H * per_slot;
    }
  }
}

template <typename T>
void encode_own_hand(const Layout &L, const Table &t, int observer, T *out) {
  std::fill(out, out + L.own_len, T(0));
  for (int i = 0; i < t.hand_n[observer]; ++i) out[i * L.card_types + t.hand[observer][i].card] = T(1);
}

// Tables are independent: big batches are split over a few host threads (joined before the call returns).
template <typename F>
void for_each_table(int n, int min_per_thread, F fn) {
  unsigned hw = std::thread::hardware_concurrency();
  int workers = std::min<int>({(int)(hw ? hw : 1), 16, n / min_per_thread});
  if (workers <= 1) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve(workers - 1);
  auto span = [&](int w) {
    const int lo = (int)((long long)n * w / workers), hi = (int)((long long)n * (w + 1) / workers);
    for (int i = lo; i < hi; ++i) fn(i);
  };
  for (int w = 1; w < workers; ++w) pool.emplace_back(span, w);
  span(0);
  for (auto &th : pool) th.join();
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

hanabi_batch_t *hanabi_batch_create(const hanabi_rules_t *rules, int32_t n_tables, const int32_t *seeds) {
  if (!rules || !seeds || n_tables <= 0) return nullptr;
  hanabi_batch *b = new (std::nothrow) hanabi_batch;
  if (!b) return nullptr;
  if (!make_layout(*rules, &b->L)) {
    delete b;
    return nullptr;
  }
  b->failed_table = -1;
  b->tables.resize((size_t)n_tables);
  for (int i = 0; i < n_tables; ++i) {
    b->tables[i].rng.seed(seeds[i]);      // std::mt19937::seed(int) as HanabiGame does (hanabi_game.cc:50)
    b->tables[i].started = false;
  }
  return b;
}

void hanabi_batch_destroy(hanabi_batch_t *b) { delete b; }

int32_t hanabi_batch_tables(const hanabi_batch_t *b) { return b ? (int32_t)b->tables.size() : HANABI_BAD_ARGUMENT; }
int32_t hanabi_batch_players(const hanabi_batch_t *b) { return b ? b->L.r.players : HANABI_BAD_ARGUMENT; }
int32_t hanabi_batch_num_moves(const hanabi_batch_t *b) { return b ? b->L.num_moves : HANABI_BAD_ARGUMENT; }
int32_t hanabi_batch_obs_len(const hanabi_batch_t *b) { return b ? b->L.obs_len : HANABI_BAD_ARGUMENT; }
int32_t hanabi_batch_own_hand_len(const hanabi_batch_t *b) { return b ? b->L.own_len : HANABI_BAD_ARGUMENT; }
int32_t hanabi_batch_failed_table(const hanabi_batch_t *b) { return b ? b->failed_table : HANABI_BAD_ARGUMENT; }

int hanabi_batch_reset(hanabi_batch_t *b, const uint8_t *choose) {
  if (!b) return HANABI_BAD_ARGUMENT;
  const int n = (int)b->tables.size();
  for (int i = 0; i < n; ++i)
    if (!choose || choose[i]) new_game(b->L, &b->tables[i]);
  return HANABI_OK;
}

int hanabi_batch_step(hanabi_batch_t *b, const int32_t *actions, float *rewards, uint8_t *status, int32_t *scores) {
  if (!b || !actions || !rewards || !status || !scores) return HANABI_BAD_ARGUMENT;
  const Layout &L = b->L;
  const int n = (int)b->tables.size();
  for (int i = 0; i < n; ++i) {
    if (!b->tables[i].started) return HANABI_NOT_STARTED;
    if (actions[i] == -1) continue;
    if (!legal(L, b->tables[i], move_of(L, actions[i]))) {
      b->failed_table = i;
      return HANABI_ILLEGAL_MOVE;
    }
  }
  for_each_table(n, 2048, [&](int i) {
    Table &t = b->tables[i];
    if (actions[i] == -1) {
      rewards[i] = 0.0f;
      status[i] = 2;
      scores[i] = score_of(L, t);
      return;
    }
    const int before = score_of(L, t);
    apply(L, &t, move_of(L, actions[i]));
    scores[i] = score_of(L, t);
    rewards[i] = (float)(scores[i] - before);
    status[i] = end_of(L, t) != kRunning ? 1 : 0;
  });
  return HANABI_OK;
}

int hanabi_batch_encode(const hanabi_batch_t *b, int share_mode, const uint8_t *active, float *obs, float *share_obs,
                        float *available, int32_t *to_move) {
  if (!b || !obs || !share_obs || !available ||
      (share_mode != HANABI_SHARE_OWN_HAND && share_mode != HANABI_SHARE_ALL_PLAYERS))
    return HANABI_BAD_ARGUMENT;
  const Layout &L = b->L;
  const int n = (int)b->tables.size(), P = L.r.players;
  const size_t obs_w = (size_t)L.obs_len + P;
  const size_t share_w = (size_t)(share_mode == HANABI_SHARE_OWN_HAND ? L.own_len + L.obs_len : P * L.obs_len) + P;
  for (int i = 0; i < n; ++i)
    if ((!active || active[i]) && !b->tables[i].started) return HANABI_NOT_STARTED;
  for_each_table(n, 128, [&](int i) {
    float *ob = obs + (size_t)i * obs_w, *sh = share_obs + (size_t)i * share_w, *av = available + (size_t)i * L.num_moves;
    if (active && !active[i]) {
      std::fill(ob, ob + obs_w, 0.0f);
      std::fill(sh, sh + share_w, 0.0f);
      std::fill(av, av + L.num_moves, 0.0f);
      if (to_move) to_move[i] = -1;
      return;
    }
    const Table &t = b->tables[i];
    const int cur = t.to_move;
    encode_view(L, t, cur, ob);
    float *turn = ob + L.obs_len;
    for (int p = 0; p < P; ++p) turn[p] = p == cur ? 1.0f : 0.0f;
    if (share_mode == HANABI_SHARE_OWN_HAND) {
      encode_own_hand(L, t, cur, sh);
      std::memcpy(sh + L.own_len, ob, obs_w * sizeof(float));           // observation | turn
    } else {
      for (int p = 0; p < P; ++p) {
        if (p == cur) std::memcpy(sh + (size_t)p * L.obs_len, ob, (size_t)L.obs_len * sizeof(float));
        else encode_view(L, t, p, sh + (size_t)p * L.obs_len);
      }
      std::memcpy(sh + (size_t)P * L.obs_len, turn, (size_t)P * sizeof(float));
    }
    for (int u = 0; u < L.num_moves; ++u) av[u] = legal(L, t, move_of(L, u)) ? 1.0f : 0.0f;
    if (to_move) to_move[i] = cur;
  });
  return HANABI_OK;
}

int hanabi_batch_player_view(const hanabi_batch_t *b, int32_t table, int32_t player, int32_t *obs, int32_t *own_hand) {
  if (!b || table < 0 || table >= (int)b->tables.size() || player < 0 || player >= b->L.r.players)
    return HANABI_BAD_ARGUMENT;
  const Table &t = b->tables[table];
  if (!t.started) return HANABI_NOT_STARTED;
  if (obs) encode_view(b->L, t, player, obs);
  if (own_hand) encode_own_hand(b->L, t, player, own_hand);
  return HANABI_OK;
}

int hanabi_batch_table_state(const hanabi_batch_t *b, int32_t table, int32_t *out) {
  if (!b || !out || table < 0 || table >= (int)b->tables.size()) return HANABI_BAD_ARGUMENT;
  const Table &t = b->tables[table];
  if (!t.started) return HANABI_NOT_STARTED;
  out[0] = t.lives;
  out[1] = t.info;
  out[2] = t.stock_total;
  out[3] = score_of(b->L, t);
  out[4] = t.to_move;
  out[5] = end_of(b->L, t);
  out[6] = t.turns_left;
  out[7] = t.discards;
  for (int c = 0; c < b->L.r.colors; ++c) out[8 + c] = t.fireworks[c];
  return HANABI_OK;
}

}  // extern "C"
