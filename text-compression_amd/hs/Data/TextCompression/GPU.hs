-- | Drop-in bodies for the Word8 / ByteString helpers of Data.BWT, Data.MTF, Data.RLE and
-- Data.FMIndex: same names, same types, the L2 `Seq (Maybe a)` kernels replaced by calls into
-- libtextcomp.so.  A maintainer re-exports these from the original modules (export lists
-- unchanged).  NOT COMPILED HERE (no GHC in the build image); kept thin and mechanical.
-- `Maybe Word8` <-> Int16, -1 = Nothing.
--
-- One process-global context (`theCtx`): a `tc_ctx` owns a HIP stream, events and the
-- device workspace, so it is created once, on first use, and every call takes it through an MVar --
-- calls on one `tc_ctx` must be serialised (include/textcomp.h, "Threading"), and the MVar is that
-- serialisation for pure code evaluated from several Haskell threads (the ...P variants, sparks).
module Data.TextCompression.GPU
  ( bytestringToBWT, bytestringFromWord8BWT, bytestringFromByteStringBWT
  , bytestringBWTToMTFB, bytestringBWTFromMTFB
  , bytestringBWTToRLEB, bytestringBWTFromRLEB
  , bytestringToBWTToFMIndexB
  , bytestringFMIndexCountS, bytestringFMIndexCountP
  , bytestringFMIndexLocateS, bytestringFMIndexLocateP
  ) where

import Control.Concurrent.MVar (MVar, newMVar, modifyMVar_, withMVar)
import Control.Monad (when)
import qualified Data.ByteString as BS
import qualified Data.ByteString.Char8 as BSC8
import qualified Data.ByteString.Unsafe as BSU
import Data.BWT.Internal (BWT (..), Suffix (..))
import Data.FMIndex.Internal (Cc (..), FMIndex (..), OccCK (..), SA (..))
import Data.Foldable (toList)
import Data.Int (Int16)
import Data.MTF.Internal (MTF (..))
import Data.RLE.Internal (RLE (..))
import qualified Data.Sequence as DS
import Data.Sequence (Seq)
import Data.Word (Word8)
import Foreign
import Foreign.C.String (peekCString)
import System.IO.Unsafe (unsafePerformIO)

import Data.TextCompression.FFI

-- | The process-global context: nullPtr until the first call creates it (never destroyed: it lives
-- as long as the process, like the RTS).
theCtx :: MVar (Ptr TcCtx)
theCtx = unsafePerformIO (newMVar nullPtr)
{-# NOINLINE theCtx #-}

-- | Run one library call on the global context, holding the MVar for its duration.
-- Two steps, so that an exception in `act` (e.g. `check` raising on the very first call) cannot lose a
-- context that was just created: `modifyMVar_` stores the new pointer BEFORE `act` runs, and `act`
-- itself runs under `withMVar` (which restores the stored pointer on any exception).
-- Discipline for callers: the MVar is not re-entrant -- never force, inside `act`, a value whose
-- evaluation calls `withCtx` again (every binding below that depends on another library call is
-- forced with `seq`/bang BEFORE `withCtx` is entered, see bytestringToBWTToFMIndexB).
withCtx :: (Ptr TcCtx -> IO a) -> IO a
withCtx act = do
  modifyMVar_ theCtx $ \c0 ->
    if c0 /= nullPtr then pure c0 else alloca $ \pp -> do
      rc <- c_tc_ctx_create 0 pp
      when (rc /= 0) (ioError (userError "textcomp: no usable HIP device"))
      peek pp
  withMVar theCtx act

-- TC_ERR_MALFORMED (-3) stands for the `error` the reference raises itself
-- (fromJust / DS.index / read); everything else is an infrastructure failure.
check :: Ptr TcCtx -> Int32 -> IO ()
check ctx rc = when (rc /= 0) $ do
  msg <- c_tc_last_error ctx >>= peekCString
  if rc == (-3) then errorWithoutStackTrace msg else ioError (userError msg)

symOf :: Maybe Word8 -> Int16
symOf = maybe (-1) fromIntegral

ofSym :: Int16 -> Maybe Word8
ofSym s = if s < 0 then Nothing else Just (fromIntegral s)

-- | BWT.hs:68-70
bytestringToBWT :: BS.ByteString -> BWT Word8
bytestringToBWT bs
  | BS.null bs = BWT DS.Empty
  | otherwise  = unsafePerformIO $ withCtx $ \ctx ->
      BSU.unsafeUseAsCStringLen bs $ \(p, n) ->
      allocaBytes (n + 1) $ \l -> alloca $ \prim -> do
        c_tc_bwt_encode ctx (castPtr p) (fromIntegral n) l prim >>= check ctx
        pr <- fromIntegral <$> peek prim
        bytes <- peekArray (n + 1) l
        pure . BWT . DS.fromList $ [ if j == pr then Nothing else Just b | (j, b) <- zip [0 :: Int ..] bytes ]

-- | BWT.hs:108-110
bytestringFromWord8BWT :: BWT Word8 -> BS.ByteString
bytestringFromWord8BWT (BWT s)
  | DS.null s = BS.empty
  | otherwise = unsafePerformIO $ withCtx $ \ctx ->
      withArrayLen (map symOf (toList s)) $ \n syms ->
      allocaBytes n $ \out -> alloca $ \nout -> do
        c_tc_bwt_decode_sym ctx syms (fromIntegral n) out nout >>= check ctx
        k <- fromIntegral <$> peek nout
        BS.packCStringLen (castPtr out, k)

-- | BWT.hs:114-116: `BS.concat . fromBWT` -- every element is a one-byte ByteString here (the shape
-- the ...B helpers produce); an element of another length has no Word8 form and is rejected.
bytestringFromByteStringBWT :: BWT BS.ByteString -> BS.ByteString
bytestringFromByteStringBWT (BWT s)
  | any bad (toList s) = errorWithoutStackTrace "bytestringFromByteStringBWT: elements must be single bytes"
  | otherwise = bytestringFromWord8BWT (BWT (fmap (fmap BS.head) s))
  where bad (Just b) = BS.length b /= 1
        bad Nothing  = False

-- | MTF.hs:117-122
bytestringBWTToMTFB :: BWT Word8 -> MTF BS.ByteString
bytestringBWTToMTFB (BWT s)
  | DS.null s = MTF (DS.Empty, DS.Empty)
  | otherwise = unsafePerformIO $ withCtx $ \ctx ->
      withArrayLen (map symOf (toList s)) $ \n syms ->
      allocaArray n $ \idx -> allocaArray 257 $ \fl -> alloca $ \sg -> do
        c_tc_mtf_encode_sym ctx syms (fromIntegral n) idx fl sg >>= check ctx
        sigma <- fromIntegral <$> peek sg
        is <- peekArray n idx
        ls <- peekArray sigma fl
        pure $ MTF (DS.fromList (map fromIntegral is), DS.fromList (map (fmap BS.singleton . ofSym) ls))

-- | MTF.hs:240-245
bytestringBWTFromMTFB :: MTF BS.ByteString -> BWT BS.ByteString
bytestringBWTFromMTFB (MTF (is, ls))
  | DS.null is || DS.null ls = BWT DS.Empty
  | otherwise = unsafePerformIO $ withCtx $ \ctx ->
      withArrayLen (map fromIntegral (toList is)) $ \n idx ->
      withArrayLen (map (symOf . fmap BS.head) (toList ls)) $ \k fl ->
      allocaArray n $ \out -> do
        c_tc_mtf_decode ctx idx (fromIntegral n) fl (fromIntegral k) out >>= check ctx
        BWT . DS.fromList . map (fmap BS.singleton . ofSym) <$> peekArray n out

-- | RLE.hs:117-123; counts rendered with `show` exactly as RLE/Internal.hs:128 does
bytestringBWTToRLEB :: BWT Word8 -> RLE BS.ByteString
bytestringBWTToRLEB (BWT s)
  | DS.null s = RLE DS.Empty
  | otherwise = unsafePerformIO $ withCtx $ \ctx ->
      withArrayLen (map symOf (toList s)) $ \n syms ->
      let cap = 2 * n + 2 in
      allocaArray cap $ \cs -> allocaArray cap $ \vs -> with (fromIntegral cap) $ \nr -> do
        c_tc_rle_encode_sym ctx syms (fromIntegral n) cs vs nr >>= check ctx
        k <- fromIntegral <$> peek nr
        counts <- peekArray k cs
        vals <- peekArray k vs
        pure . RLE . DS.fromList . concat $
          [ [Just (BSC8.pack (show c)), fmap BS.singleton (ofSym v)] | (c, v) <- zip counts vals ]

-- | RLE.hs:237-241 (pairs; `(Just _, Nothing)` is one Nothing; an odd tail is ignored)
bytestringBWTFromRLEB :: RLE BS.ByteString -> BWT BS.ByteString
bytestringBWTFromRLEB (RLE s)
  | DS.null s = BWT DS.Empty
  | otherwise = unsafePerformIO $ withCtx $ \ctx -> do
      let prs = pairs (toList s)
          cnt (Just c, Just _) = max 0 (read (BSC8.unpack c)) :: Int   -- `read` throws as the reference does
          cnt _ = 1
          sym (_, Nothing) = -1
          sym (_, Just b) = fromIntegral (BS.head b)
          total = sum [ if sym p < 0 then 1 else cnt p | p <- prs ]
      withArrayLen (map (fromIntegral . cnt) prs) $ \k cs ->
        withArray (map sym prs) $ \vs ->
        allocaArray (total + 1) $ \out -> with (fromIntegral (total + 1)) $ \nn -> do
          c_tc_rle_decode ctx cs vs (fromIntegral k) out nn >>= check ctx
          m <- fromIntegral <$> peek nn
          BWT . DS.fromList . map (fmap BS.singleton . ofSym) <$> peekArray m out
  where pairs (a : b : r) = (a, b) : pairs r
        pairs _ = []

-- | FMIndex.hs:108-111 (via bytestringBWTToFMIndexB :162-183): the FMIndex VALUE (Cc, OccCK, SA) -- sigma
-- pairs, sigma x N triples and N suffix records, a shape for small inputs only (the reference builds it
-- through the O(n^2) rotation matrix).  The device supplies what the path computes -- last column, C[c],
-- suffix array -- and the Seq-of-tuples shape is laid out here exactly as seqToCc / seqToOccCK
-- (FMIndex/Internal.hs:195-316: rows for the present symbols, Nothing first; Occ inclusive of k, k from 1)
-- and createSuffixArray (BWT/Internal.hs:110-134: 1-based rank and start, Nothing for the empty suffix) do.
bytestringToBWTToFMIndexB :: BS.ByteString -> FMIndex BS.ByteString
bytestringToBWTToFMIndexB bs
  | BS.null bs = FMIndex (Cc DS.Empty, OccCK DS.Empty, SA DS.Empty)
  | otherwise = unsafePerformIO $ do
      let BWT l = bytestringToBWT bs
          lcol  = fmap (fmap BS.singleton) l                      -- Seq (Maybe ByteString)
          n     = BS.length bs
      (syms, cvals, sa) <- withCtx $ \ctx ->
        BSU.unsafeUseAsCStringLen bs $ \(p, _) -> alloca $ \ph -> do
          c_tc_fm_build ctx (castPtr p) (fromIntegral n) ph >>= check ctx
          fm <- peek ph
          r <- alloca $ \pn -> alloca $ \psg -> allocaArray 257 $ \cs -> allocaArray 257 $ \cv -> alloca $ \pp -> do
                 _ <- c_tc_fm_info fm pn psg cs cv pp
                 sg <- fromIntegral <$> peek psg
                 (,) <$> peekArray sg cs <*> peekArray sg cv
          c_tc_fm_free fm
          sarr <- allocaArray (n + 1) $ \ps -> do
                    c_tc_suffix_array ctx (castPtr p) (fromIntegral n) ps >>= check ctx
                    peekArray (n + 1) ps
          pure (fst r, snd r, sarr)
      let el s   = fmap BS.singleton (ofSym s)
          cc     = DS.fromList [ (fromIntegral v, el s) | (s, v) <- zip syms cvals ]
          row c  = DS.fromList . snd $
                     foldl (\(run, acc) (k, x) -> let run' = if x == c then run + 1 else run
                                                  in (run', acc ++ [(k, run', x)]))
                           (0 :: Int, []) (zip [1 ..] (toList lcol))
          occck  = DS.fromList [ (el s, row (el s)) | s <- syms ]
          sufOf i = let t = BS.drop i bs
                    in if BS.null t then Nothing else Just (DS.fromList (map BS.singleton (BS.unpack t)))
          sarec  = DS.fromList [ Suffix { suffixindex = j, suffixstartpos = fromIntegral i + 1, suffix = sufOf (fromIntegral i) }
                               | (j, i) <- zip [1 ..] sa ]
      pure (FMIndex (Cc cc, OccCK occck, SA sarec))

-- | FMIndex.hs:362-379: one batched device call; 0 stands for Nothing; order preserved
bytestringFMIndexCountS :: [BS.ByteString] -> BS.ByteString -> Seq (BS.ByteString, Maybe Int)
bytestringFMIndexCountS pats input
  | null pats || BS.null input = DS.Empty
  | otherwise = unsafePerformIO $ withCtx $ \ctx ->
      BSU.unsafeUseAsCStringLen input $ \(p, n) -> alloca $ \ph -> do
        c_tc_fm_build ctx (castPtr p) (fromIntegral n) ph >>= check ctx
        fm <- peek ph
        let flat = BS.concat pats `BS.snoc` 0
            offs = scanl (+) 0 (map (fromIntegral . BS.length) pats)
        r <- BSU.unsafeUseAsCString flat $ \fp -> withArray offs $ \op ->
               allocaArray (length pats) $ \out -> do
                 rc <- c_tc_fm_count ctx fm (castPtr fp) op (fromIntegral (length pats)) out
                 c_tc_fm_free fm
                 check ctx rc
                 peekArray (length pats) out
        pure . DS.fromList $ [ (q, if c == 0 then Nothing else Just (fromIntegral c)) | (q, c) <- zip pats r ]

-- | FMIndex.hs:475-497: 1-based text positions `suffixstartpos (sa[x-1])`, in suffix-array order (not
-- sorted); no occurrence (or an empty pattern) is the empty Seq.  One batched device call: tc_fm_locate
-- returns the hits of all patterns back to back with an offset array; a first call sizes the buffer.
bytestringFMIndexLocateS :: [BS.ByteString] -> BS.ByteString -> Seq (BS.ByteString, Seq (Maybe Int))
bytestringFMIndexLocateS pats input
  | null pats || BS.null input = DS.Empty
  | otherwise = unsafePerformIO $ withCtx $ \ctx ->
      BSU.unsafeUseAsCStringLen input $ \(p, n) -> alloca $ \ph -> do
        c_tc_fm_build ctx (castPtr p) (fromIntegral n) ph >>= check ctx
        fm <- peek ph
        let np   = length pats
            flat = BS.concat pats `BS.snoc` 0
            offs = scanl (+) 0 (map (fromIntegral . BS.length) pats)
            -- one attempt with room for `cap` hits: Left needed | Right (offsets, hits)
            attempt :: Int -> IO (Either Int ([Int], [Int]))
            attempt cap =
              BSU.unsafeUseAsCString flat $ \fp -> withArray offs $ \op ->
              allocaArray (np + 1) $ \ho -> allocaArray (max 1 cap) $ \hits ->
              with (fromIntegral cap) $ \nh -> do
                rc <- c_tc_fm_locate ctx fm (castPtr fp) op (fromIntegral np) ho hits nh
                total <- fromIntegral <$> peek nh
                if rc == (-2)                                   -- TC_ERR_CAPACITY: *nhits = hits needed
                  then pure (Left total)
                  else do
                    when (rc /= 0) (c_tc_fm_free fm)
                    check ctx rc
                    os <- map fromIntegral <$> peekArray (np + 1) ho
                    hs <- map fromIntegral <$> peekArray total hits
                    pure (Right (os, hs))
        r1 <- attempt (16 * np)
        (os, hs) <- case r1 of
          Right v   -> pure v
          Left need -> attempt need >>= either (\_ -> c_tc_fm_free fm >> ioError (userError "tc_fm_locate: capacity")) pure
        c_tc_fm_free fm
        let slices = [ take (e - a) (drop a hs) | (a, e) <- zip os (tail os) ]
        pure . DS.fromList $ [ (q, DS.fromList (map Just h)) | (q, h) <- zip pats slices ]

-- | FMIndex.hs:538-563: the parListChunk spark pool is the one batched launch above
bytestringFMIndexLocateP :: [BS.ByteString] -> BS.ByteString -> IO (Seq (BS.ByteString, Seq (Maybe Int)))
bytestringFMIndexLocateP pats input = pure (bytestringFMIndexLocateS pats input)

-- | FMIndex.hs:411-432: the parListChunk spark pool is the one batched launch above
bytestringFMIndexCountP :: [BS.ByteString] -> BS.ByteString -> IO (Seq (BS.ByteString, Maybe Int))
bytestringFMIndexCountP pats input = pure (bytestringFMIndexCountS pats input)
